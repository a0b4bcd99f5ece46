"""Row-sharded device path on ONE GPU: (a) world = 1 with an identity all-reduce exercises the
statistics / all-reduce / draw / apply kernels; (b) two shards in two lock-stepped threads whose
all-reduce callback sums the two device buffers -- the same call sequence a 2-GPU RCCL run makes.
Both must reproduce the oracle's unsharded chain (regression, same seed)."""
import threading

import numpy as np
import pytest
import scipy.sparse as sps

from . import datasets as ds

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _small_tables_take_the_persistent_sweep(monkeypatch):
    """The persistent sweep is the default from 2^20 rows on; these tests mean to cover it on small tables whatever mode the suite
    runs in (conftest sets the same for the checker mode; MYFM_TEST_PRODUCTION=1 does not)."""
    monkeypatch.setenv("MFM_RES_MIN_ROWS", "0")


def _config(gi, n_iter=8, task=None):
    from myfm_amd import _myfm

    b = _myfm.ConfigBuilder()
    b.set_alpha_0(1.0).set_beta_0(1.0).set_gamma_0(1.0).set_mu_0(0.0).set_reg_0(1.0)
    b.set_group_index([int(g) for g in gi]).set_n_iter(n_iter).set_n_kept_samples(0)
    b.set_task_type(task if task is not None else _myfm.TaskType.REGRESSION)
    return b.build()


def _designs(myfm_mod):
    X, y, shapes = ds.onehot_mf(30000, 400, 60, seed=3)
    Xd = sps.hstack([X, sps.csr_matrix(np.random.default_rng(1).normal(size=(30000, 1)))]).tocsr()
    gid = np.concatenate([ds.group_index_from_shapes(shapes), [2]])
    main, X_flat, blocks, yb, bshapes = ds.multihot_block_design(n_train=600)
    return {
        "onehot_plus_dense": (Xd, y, [], gid, 4),
        "relation_blocks": (main, yb, blocks, ds.group_index_from_shapes(bshapes), 3),
    }


@pytest.mark.parametrize("name", ["onehot_plus_dense", "relation_blocks"])
def test_world1_identity_allreduce(oracle, name):
    import myfm_amd
    from myfm_amd import _myfm

    X, y, blocks, gi, K = _designs(myfm_amd)[name]
    calls = []
    rbs = [_myfm.RelationBlock([int(v) for v in m], b) for m, b in blocks]
    from myfm_amd import _capi

    s = _myfm.GibbsSession(K, 0.1, X, rbs, y, 42, _config(gi), allreduce=lambda p, c: calls.append(c), n_total_rows=X.shape[0],
                           main_levels=_capi.column_levels(X)[0])
    t = oracle.OracleTrainer(X, y, blocks, rank=K, group_index=gi)
    for it in range(4):
        s.step()
        t.step()
    np.testing.assert_allclose(s.fm.V, t.fm()[2], rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(s.fm.w, t.fm()[1], rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(s.residual(), t.e(X.shape[0]), rtol=1e-7, atol=1e-7)
    assert len(calls) > 4 * (K + 1)  # one collective per level of every sweep (+ reductions)


class Lockstep:
    """sums the buffers of `world` sessions living in one process (stand-in for RCCL on one GPU)"""

    def __init__(self, world):
        import torch

        self.torch = torch
        self.world = world
        self.bar = threading.Barrier(world)
        self.bufs = [None] * world
        self.counts = [0] * world

    def callback(self, rank):
        from myfm_amd.distributed import _DevView

        def cb(ptr, count):
            torch = self.torch
            torch.cuda.synchronize()
            self.bufs[rank] = torch.as_tensor(_DevView(ptr, count), device="cuda")
            self.counts[rank] += 1
            self.bar.wait()
            total = self.bufs[0].clone()
            for b in self.bufs[1:]:
                assert b.shape == total.shape
                total += b
            torch.cuda.synchronize()
            self.bar.wait()
            self.bufs[rank].copy_(total)
            torch.cuda.synchronize()
            self.bar.wait()

        return cb


@pytest.mark.parametrize("name,unsync", [("onehot_plus_dense", "inverse"), ("relation_blocks", "inverse"),
                                         ("relation_blocks", "stream"), ("relation_blocks", "split")])
def test_two_shards_lockstep(oracle, monkeypatch, name, unsync):
    # unsync: form of the relation blocks' statistics / un-sync pass on every shard (inverse map | streaming with the sums in
    # LDS | streaming un-sync + read-only statistics); the latter two also fold each block's re-sync into the next pass
    monkeypatch.setenv("MFM_UNSYNC_STREAM_FORCE", "1" if unsync == "stream" else "0")
    monkeypatch.setenv("MFM_UNSYNC_SPLIT_FORCE", "1" if unsync == "split" else "0")
    import myfm_amd
    from myfm_amd import _myfm
    from myfm_amd.distributed import shard_rows

    X, y, blocks, gi, K = _designs(myfm_amd)[name]
    from myfm_amd import _capi

    world = 2
    ls = Lockstep(world)
    levels = _capi.column_levels(X)[0]  # schedule of the GLOBAL design, identical on every rank
    out, errs = {}, []

    def run(rank):
        try:
            Xl, yl, rel, lo, n = shard_rows(X, y, blocks, rank, world)
            rbs = [_myfm.RelationBlock([int(v) for v in m], b) for m, b in rel]
            s = _myfm.GibbsSession(K, 0.1, Xl, rbs, yl, 42, _config(gi), allreduce=ls.callback(rank), n_total_rows=n,
                                   row_offset=lo, main_levels=levels, shard_rank=rank, shard_world=world)
            for it in range(4):
                s.step()
            out[rank] = (s.fm.w0, np.asarray(s.fm.w), np.asarray(s.fm.V), s.residual(), lo, float(s.hyper.alpha))
        except BaseException as ex:  # noqa
            errs.append(ex)
            ls.bar.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join(timeout=300)
    assert not errs, errs
    t = oracle.OracleTrainer(X, y, blocks, rank=K, group_index=gi)
    for it in range(4):
        t.step()
    w0, w, V = t.fm()
    e = t.e(X.shape[0])
    for rank in range(world):
        gw0, gw, gV, ge, lo, galpha = out[rank]
        assert abs(gw0 - w0) < 1e-7
        np.testing.assert_allclose(gw, w, rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(gV, V, rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(ge, e[lo:lo + ge.shape[0]], rtol=1e-7, atol=1e-7)
        assert abs(galpha - t.hyper()["alpha"]) < 1e-7 * galpha
    assert ls.counts[0] == ls.counts[1] > 0


@pytest.mark.parametrize("world,values,n_fields", [(2, False, 2), (3, True, 2), (2, True, 3), (3, False, 4)])
def test_sharded_fused_tile_path(oracle, world, values, n_fields, monkeypatch):
    """user-sorted two-field table, row-sharded: first-level columns complete on one rank are swept locally inside
    the fused tile pass, the users straddling a rank boundary (and one longer than a tile) go through the
    all-reduced path, the model is synchronised after the sweep -- must reproduce the unsharded oracle chain"""
    from myfm_amd import _capi, _myfm
    from myfm_amd.distributed import shard_rows

    monkeypatch.setenv("MFM_SCATTER_MIN_NNZ", "1000")
    n = 90001
    X, y, shapes = ds.onehot_mf(n, 80, 70, seed=9, sort_by_user=True)
    for extra in range(n_fields - 2):  # more one-hot fields: the multi-level form of the fused pass, sharded
        k = 11 + 6 * extra
        ctx = np.random.default_rng(50 + extra).integers(0, k, size=n)
        X = sps.hstack([X, sps.csr_matrix((np.ones(n), ctx, np.arange(n + 1)), shape=(n, k))]).tocsr()
        X.sort_indices()
        shapes = shapes + [k]
    if values:
        X = X.copy()
        X.data = np.where(np.arange(X.nnz) % 3 == 0, 0.5, 1.5)
    assert np.diff(X.tocsc().indptr)[:80].max() > 4096
    gi = ds.group_index_from_shapes(shapes)
    K = 3
    ls = Lockstep(world)
    levels = _capi.column_levels(X)[0]
    out, errs = {}, []

    def run(rank):
        try:
            Xl, yl, rel, lo, ntot = shard_rows(X, y, [], rank, world)
            s = _myfm.GibbsSession(K, 0.1, Xl, [], yl, 42, _config(gi), allreduce=ls.callback(rank), n_total_rows=ntot,
                                   row_offset=lo, main_levels=levels, shard_rank=rank, shard_world=world)
            flags = s.plan_flags()
            for it in range(3):
                s.step()
            out[rank] = (s.fm.w0, np.asarray(s.fm.w), np.asarray(s.fm.V), s.residual(), lo, float(s.hyper.alpha), flags)
        except BaseException as ex:  # noqa
            errs.append(ex)
            ls.bar.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join(timeout=300)
    assert not errs, errs
    t = oracle.OracleTrainer(X, y, rank=K, group_index=gi)
    for it in range(3):
        t.step()
    w0, w, V = t.fm()
    e = t.e(n)
    for rank in range(world):
        gw0, gw, gV, ge, lo, galpha, flags = out[rank]
        assert flags & 64, flags  # the fused sharded path was taken
        assert abs(gw0 - w0) < 1e-7
        np.testing.assert_allclose(gw, w, rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(gV, V, rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(ge, e[lo:lo + ge.shape[0]], rtol=1e-7, atol=1e-7)
        assert abs(galpha - t.hyper()["alpha"]) < 1e-7 * galpha


@pytest.mark.parametrize("seed,world", [(1, 2), (6, 3), (9, 2), (4, 2), (0, 3)])
def test_random_designs_sharded(oracle, seed, world, monkeypatch):
    """the seeded random one-hot designs of test_gpu_random_designs.py, row-sharded over lock-stepped ranks (two-field
    sorted tables take the fused sharded path, the others the generic per-level all-reduce path)"""
    from myfm_amd import _capi, _myfm
    from myfm_amd.distributed import shard_rows

    from .test_gpu_random_designs import _random_design

    X, y, gi, K, env = _random_design(seed)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ls = Lockstep(world)
    levels = _capi.column_levels(X)[0]
    out, errs = {}, []

    def run(rank):
        try:
            Xl, yl, rel, lo, ntot = shard_rows(X, y, [], rank, world)
            s = _myfm.GibbsSession(K, 0.1, Xl, [], yl, 42, _config(gi), allreduce=ls.callback(rank), n_total_rows=ntot,
                                   row_offset=lo, main_levels=levels, shard_rank=rank, shard_world=world)
            for it in range(3):
                s.step()
            out[rank] = (s.fm.w0, np.asarray(s.fm.w), np.asarray(s.fm.V), s.residual(), lo, s.plan_flags())
        except BaseException as ex:  # noqa
            errs.append(ex)
            ls.bar.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join(timeout=300)
    assert not errs, errs
    t = oracle.OracleTrainer(X, y, rank=K, group_index=gi)
    for it in range(3):
        t.step()
    w0, w, V = t.fm()
    e = t.e(X.shape[0])
    for rank in range(world):
        gw0, gw, gV, ge, lo, flags = out[rank]
        assert abs(gw0 - w0) < 1e-7, flags
        np.testing.assert_allclose(gV, V, rtol=1e-7, atol=1e-8, err_msg=str(flags))
        np.testing.assert_allclose(gw, w, rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(ge, e[lo:lo + ge.shape[0]], rtol=1e-7, atol=1e-7)


def test_world1_native_rccl(oracle):
    # the library's own RCCL all-reduce (mfm_comm_init: ncclCommInitRank + ncclAllReduce on the ctx stream), world = 1:
    # the communicator, the per-level call sequence and the fused sharded tile path, against the oracle's chain
    from myfm_amd import _capi, _myfm

    X, y, shapes = ds.onehot_mf(30000, 400, 60, seed=3)
    gi = ds.group_index_from_shapes(shapes)
    cid = _myfm.comm_unique_id()
    assert len(cid) == 128
    s = _myfm.GibbsSession(4, 0.1, X, [], y, 42, _config(gi), n_total_rows=X.shape[0], main_levels=_capi.column_levels(X)[0],
                           comm_id=cid, shard_rank=0, shard_world=1)
    t = oracle.OracleTrainer(X, y, rank=4, group_index=gi)
    for it in range(4):
        s.step()
        t.step()
    np.testing.assert_allclose(s.fm.V, t.fm()[2], rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(s.residual(), t.e(X.shape[0]), rtol=1e-7, atol=1e-7)
    calls, doubles = s.comm_stats()
    assert calls > 4 * 5 and doubles > 0
    assert s.plan_flags() & 8


def test_multi_gpu_sharded_fit():
    # real multi-GPU run (one process per GPU, RCCL over xGMI): needs >= 2 visible GPUs, skipped otherwise
    import os
    import subprocess
    import sys

    from myfm_amd import _myfm

    n = _myfm.device_count()
    if n < 2:
        pytest.skip("needs >= 2 GPUs (this box has %d)" % n)
    world = 2 if n < 4 else 4
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", "29631", os.path.join(root, "tests", "mp_fit_worker.py")]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "mp_fit_worker ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_two_processes_one_gpu_sharded_fit():
    # the multi-PROCESS flow of the row-sharded fit() on a 1-GPU box: two ranks under torch.distributed.run, both on
    # device 0, process group over gloo, the library's all-reduces through the torch.distributed callback; each rank
    # reproduces the oracle's unsharded chain and the replicas agree bit for bit
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29641", os.path.join(root, "tests", "mp_fit_worker.py")]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MP_FIT_ONE_GPU="1")
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "mp_fit_worker ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.parametrize("peer_model", ["0", "1"])
def test_two_processes_one_gpu_persistent_sweep(peer_model):
    # the row-sharded persistent sweep with its ranks in two PROCESSES (one GPU: both on device 0, each with a share of the CUs):
    # IPC handles of the exchange buffers over torch.distributed, the launches of the two processes side by side, item sums
    # exchanged inside the launch; oracle chain on every rank, replicas bit-identical (tests/mp_peer_worker.py)
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29643", os.path.join(root, "tests", "mp_peer_worker.py")]
    # MYFM_PEER_MODEL=0 (the default): the ranks' item sums meet inside the launch, the first-level coefficients are made equal by the
    # model all-reduce after it; =1: a coefficient is also written into every replica where it is drawn (no model all-reduce)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MFM_RES_NO_PROCESS_LOCK="1", MFM_RES_CUS="100", MFM_RES_MIN_ROWS="0",
               MFM_SCATTER_MIN_NNZ="1000", MYFM_PEER_MODEL=peer_model)
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "mp_peer_worker ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_empty_shard_does_not_hang_or_corrupt(oracle, monkeypatch):
    # a rank without rows must take part in every collective of mfm_finalize and of the sweeps, and the owner of
    # the replicated columns is rank 0 of the communicator, not "the rank whose first row is global row 0"
    from myfm_amd import _capi, _myfm

    monkeypatch.setenv("MFM_SCATTER_MIN_NNZ", "1000")
    n = 30000
    X, y, shapes = ds.onehot_mf(n, 50, 40, seed=12, sort_by_user=True)
    gi = ds.group_index_from_shapes(shapes)
    K, world = 3, 3
    cuts = [0, 0, 14000, n]  # rank 0 holds no rows (and rank 1 starts at global row 0 too)
    ls = Lockstep(world)
    levels = _capi.column_levels(X)[0]
    out, errs = {}, []

    def run(rank):
        try:
            lo, hi = cuts[rank], cuts[rank + 1]
            s = _myfm.GibbsSession(K, 0.1, X[lo:hi], [], y[lo:hi], 42, _config(gi), allreduce=ls.callback(rank), n_total_rows=n,
                                   row_offset=lo, main_levels=levels, shard_rank=rank, shard_world=world)
            for it in range(3):
                s.step()
            out[rank] = (np.asarray(s.fm.w), np.asarray(s.fm.V), s.residual(), lo)
        except BaseException as ex:  # noqa
            errs.append(ex)
            ls.bar.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join(timeout=300)
    assert not errs, errs
    t = oracle.OracleTrainer(X, y, rank=K, group_index=gi)
    for it in range(3):
        t.step()
    _, w, V = t.fm()
    for rank in range(world):
        gw, gV, ge, lo = out[rank]
        np.testing.assert_allclose(gw, w, rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(gV, V, rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(ge, t.e(n)[lo:lo + ge.shape[0]], rtol=1e-7, atol=1e-7)


@pytest.mark.parametrize("world,values", [(2, False), (3, True)])
def test_sharded_two_field_pass(oracle, world, values, monkeypatch):
    """shards cut BETWEEN users (myfm_amd.distributed.shard_cuts: what fit() and bench.py do): no first-level column
    straddles a rank, the q-cache-free two-field pass runs on every rank with one all-reduce of the item statistics per
    factor; one user is longer than a tile (solo tiles + finish pass)"""
    from myfm_amd import _capi, _myfm
    from myfm_amd.distributed import shard_cuts

    monkeypatch.setenv("MFM_SCATTER_MIN_NNZ", "1000")
    n = 90001
    X, y, shapes = ds.onehot_mf(n, 80, 70, seed=9, sort_by_user=True)
    if values:
        X = X.copy()
        X.data = np.where(np.arange(X.nnz) % 3 == 0, 0.5, 1.5)
    gi = ds.group_index_from_shapes(shapes)
    K = 3
    cuts = shard_cuts(X.indices[X.indptr[:-1]], world)
    ls = Lockstep(world)
    levels = _capi.column_levels(X)[0]
    out, errs = {}, []

    def run(rank):
        try:
            lo, hi = cuts[rank], cuts[rank + 1]
            s = _myfm.GibbsSession(K, 0.1, X[lo:hi], [], y[lo:hi], 42, _config(gi), allreduce=ls.callback(rank), n_total_rows=n,
                                   row_offset=lo, main_levels=levels, shard_rank=rank, shard_world=world)
            flags = s.plan_flags()
            for it in range(3):
                s.step()
            out[rank] = (s.fm.w0, np.asarray(s.fm.w), np.asarray(s.fm.V), s.residual(), lo, flags)
        except BaseException as ex:  # noqa
            errs.append(ex)
            ls.bar.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join(timeout=300)
    assert not errs, errs
    t = oracle.OracleTrainer(X, y, rank=K, group_index=gi)
    for it in range(3):
        t.step()
    w0, w, V = t.fm()
    e = t.e(n)
    for rank in range(world):
        gw0, gw, gV, ge, lo, flags = out[rank]
        assert flags & 64 and flags & 128, flags  # sharded fused path + two-field pass
        assert abs(gw0 - w0) < 1e-7
        np.testing.assert_allclose(gw, w, rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(gV, V, rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(ge, e[lo:lo + ge.shape[0]], rtol=1e-7, atol=1e-7)


def test_sharded_persistent_sweep_refused_on_one_rank_falls_back_everywhere(oracle, monkeypatch):
    """Two ranks on one GPU that each ask for 200 of its 256 CUs: the second claim is refused, mfm_finalize's agreement sends
    BOTH ranks to the per-factor passes (no layout waiting for peers anywhere), and the chain is the oracle's."""
    from myfm_amd import _capi, _myfm
    from myfm_amd.distributed import shard_cuts

    monkeypatch.setenv("MFM_SCATTER_MIN_NNZ", "1000")
    monkeypatch.setenv("MFM_RES_CUS", "200")
    monkeypatch.setenv("MFM_RES_WGS", "200")  # (as many workgroups as allowed, so that the claims collide)
    n, world, K = 120001, 2, 3
    X, y, shapes = ds.onehot_mf(n, 300, 170, seed=13, sort_by_user=True)
    gi = ds.group_index_from_shapes(shapes)
    cuts = shard_cuts(X.indices[X.indptr[:-1]], world)
    ls = Lockstep(world)
    levels = _capi.column_levels(X)[0]
    out, errs = {}, []

    def run(rank):
        try:
            lo, hi = cuts[rank], cuts[rank + 1]
            s = _myfm.GibbsSession(K, 0.1, X[lo:hi], [], y[lo:hi], 42, _config(gi), allreduce=ls.callback(rank), n_total_rows=n,
                                   row_offset=lo, main_levels=levels, shard_rank=rank, shard_world=world)
            pending = s.peer_info()[0]
            for it in range(2):
                s.step()
            out[rank] = (pending, s.plan_flags(), np.asarray(s.fm.V))
        except BaseException as ex:  # noqa
            errs.append(ex)
            ls.bar.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join(timeout=300)
    assert not errs, errs
    t = oracle.OracleTrainer(X, y, rank=K, group_index=gi)
    for it in range(2):
        t.step()
    for rank in range(world):
        pending, flags, V = out[rank]
        assert not pending and not (flags & 256) and flags & 128, (pending, flags)
        np.testing.assert_allclose(V, t.fm()[2], rtol=1e-7, atol=1e-8)


@pytest.mark.parametrize("model", [True, False])
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_persistent_sweep_in_launch_exchange(oracle, world, model, monkeypatch):
    """The persistent sweep row-sharded (mfm_res.hpp XCH): every rank keeps the residual of ITS rows on chip for all
    factors, the ranks' item sums meet INSIDE the launch -- each workgroup writes its slice's sums into every rank's exchange
    buffer, the last workgroup of a rank raises the rank's flag on the peers, everybody adds the ranks' sums in rank order -- so
    there is no collective between the sweeps of an iteration (one model synchronisation after the launch). Here the ranks
    are sessions of one process on ONE GPU, each claiming a share of the CUs, running their launches side by side; the peers'
    buffers are plain device pointers (mfm_peer_set). model: the peers' w / V are known as well (mfm_peer_set_model) -- a user's
    coefficient is written to every replica where it is drawn, and the only collective left in an iteration is the residual sum.
    Must reproduce the unsharded oracle chain; replicas identical."""
    from myfm_amd import _capi, _myfm
    from myfm_amd.distributed import shard_cuts

    monkeypatch.setenv("MFM_SCATTER_MIN_NNZ", "1000")
    monkeypatch.setenv("MFM_RES_CUS", str(240 // world))
    n = 120001
    X, y, shapes = ds.onehot_mf(n, 300, 170, seed=13, sort_by_user=True)
    gi = ds.group_index_from_shapes(shapes)
    K = 3
    cuts = shard_cuts(X.indices[X.indptr[:-1]], world)
    ls = Lockstep(world)
    levels = _capi.column_levels(X)[0]
    out, errs, peers = {}, [], {}
    meet = threading.Barrier(world)

    def run(rank):
        try:
            lo, hi = cuts[rank], cuts[rank + 1]
            s = _myfm.GibbsSession(K, 0.1, X[lo:hi], [], y[lo:hi], 42, _config(gi), allreduce=ls.callback(rank), n_total_rows=n,
                                   row_offset=lo, main_levels=levels, shard_rank=rank, shard_world=world)
            pending, sum_p, flag_p, sum_b, flag_b = s.peer_info()
            assert pending and sum_p and flag_p and sum_b > 0, (pending, sum_p, flag_p)
            assert not (s.plan_flags() & 256)  # not live before every rank knows every rank's buffers
            peers[rank] = (sum_p, flag_p) + tuple(s.peer_model_info())
            meet.wait()
            s.peer_set(world, rank, [peers[r][0] for r in range(world)], [peers[r][1] for r in range(world)])
            if model:
                s.peer_set_model(world, rank, [peers[r][2] for r in range(world)], [peers[r][3] for r in range(world)])
            flags = s.plan_flags()
            calls0 = ls.counts[rank]
            for it in range(3):
                s.step()
            out[rank] = (s.fm.w0, np.asarray(s.fm.w), np.asarray(s.fm.V), s.residual(), lo, flags, ls.counts[rank] - calls0)
        except BaseException as ex:  # noqa
            errs.append(ex)
            ls.bar.abort()
            meet.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join(timeout=300)
    assert not errs, errs
    t = oracle.OracleTrainer(X, y, rank=K, group_index=gi)
    for it in range(3):
        t.step()
    w0, w, V = t.fm()
    e = t.e(n)
    for rank in range(world):
        gw0, gw, gV, ge, lo, flags, calls = out[rank]
        assert flags & 256 and flags & 8, flags  # persistent sweep, row-sharded
        # per iteration: the residual sums, and without the peers' replicas the w and V synchronisation -- nothing per factor
        assert calls <= (3 * 2 if model else 3 * 6), calls
        assert abs(gw0 - w0) < 1e-7
        np.testing.assert_allclose(gw, w, rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(gV, V, rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(ge, e[lo:lo + ge.shape[0]], rtol=1e-7, atol=1e-7)
        assert np.array_equal(gV, out[0][2]) and np.array_equal(gw, out[0][1])


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_sharded_persistent_sweep_awkward_shards(oracle, seed, monkeypatch):
    """The same path on tables with columns that never occur (in both fields: drawn from the prior by every rank alike), items
    that have rows on one rank only, and uneven shards: oracle chain, identical replicas."""
    from myfm_amd import _capi, _myfm

    from .test_gpu_planner_fuzz import _two_field

    monkeypatch.setenv("MFM_SCATTER_MIN_NNZ", "1000")
    world = 3
    monkeypatch.setenv("MFM_RES_CUS", str(240 // world))
    rng = np.random.default_rng(50 + seed)
    n = int(rng.integers(60000, 140000))
    X, y, shapes = _two_field(rng, n, int(rng.integers(40, 400)), int(rng.integers(30, 300)), float(rng.uniform(0.3, 1.4)), 0.3)
    gi = ds.group_index_from_shapes(shapes)
    K = 2
    first = X.indices[X.indptr[:-1]]
    users = np.unique(first)
    # rank 0: a small shard (the first sixth of the users); the rest split in two at a user boundary
    b0 = int(np.searchsorted(first, users[max(3, len(users) // 6)]))
    b1 = int(np.searchsorted(first, users[(len(users) + 3) // 2]))
    cuts = [0, b0, max(b1, b0 + 1), n]
    ls = Lockstep(world)
    levels = _capi.column_levels(X)[0]
    out, errs, peers = {}, [], {}
    meet = threading.Barrier(world)

    def run(rank):
        try:
            lo, hi = cuts[rank], cuts[rank + 1]
            s = _myfm.GibbsSession(K, 0.1, X[lo:hi], [], y[lo:hi], 42, _config(gi), allreduce=ls.callback(rank), n_total_rows=n,
                                   row_offset=lo, main_levels=levels, shard_rank=rank, shard_world=world)
            info = s.peer_info()
            peers[rank] = (info[0], info[1], info[2]) + tuple(s.peer_model_info())
            meet.wait()
            assert all(peers[r][0] for r in range(world)) == bool(info[0])  # the same verdict on every rank
            if info[0]:
                s.peer_set(world, rank, [peers[r][1] for r in range(world)], [peers[r][2] for r in range(world)])
                s.peer_set_model(world, rank, [peers[r][3] for r in range(world)], [peers[r][4] for r in range(world)])
            for it in range(3):
                s.step()
            out[rank] = (bool(info[0]), s.plan_flags(), np.asarray(s.fm.w), np.asarray(s.fm.V), s.residual(), lo)
        except BaseException as ex:  # noqa
            errs.append(ex)
            ls.bar.abort()
            meet.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join(timeout=300)
    assert not errs, errs
    t = oracle.OracleTrainer(X, y, rank=K, group_index=gi)
    for it in range(3):
        t.step()
    w0, w, V = t.fm()
    e = t.e(n)
    assert out[0][0], "the layout was refused: the test does not reach the exchange"
    for rank in range(world):
        live, flags, gw, gV, ge, lo = out[rank]
        assert flags & 256
        np.testing.assert_allclose(gw, w, rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(gV, V, rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(ge, e[lo:lo + ge.shape[0]], rtol=1e-7, atol=1e-7)
        assert np.array_equal(gV, out[0][3]) and np.array_equal(gw, out[0][2])


@pytest.mark.parametrize("world,shape", [(2, "u_i_ctx"), (3, "u_i_ctx"), (2, "no_item"), (3, "three_fields")])
def test_cell_path_row_sharded(oracle, monkeypatch, world, shape):
    """Index-tuple designs (mfm_cell.hpp) row-sharded over `world` lock-stepped ranks on one GPU: every rank runs the cell
    passes over its own rows (users straddle the rank boundaries), a field's sums go through one dense array that is
    all-reduced before the replicated draw / feature sweep (the all-reduce of the per-block sufficient statistics).
    Must reproduce the unsharded oracle chain; the ranks' models must be identical."""
    from myfm_amd import _capi, _myfm
    from myfm_amd.distributed import shard_rows

    from .test_gpu_cell import SHAPES

    monkeypatch.setenv("MFM_CELL_MIN_ROWS", "0")
    monkeypatch.setenv("MFM_CELL_GROUPS", "7")
    main, blocks, y, shapes = ds.tuple_design(**SHAPES[shape])
    gi = ds.group_index_from_shapes(shapes)
    K, n_iter = 3, 3
    ls = Lockstep(world)
    levels = _capi.column_levels(main)[0]
    out, errs = {}, []

    def run(rank):
        try:
            Xl, yl, rel, lo, n = shard_rows(main, y, blocks, rank, world)
            rbs = [_myfm.RelationBlock(np.asarray(m, dtype=np.int64), b) for m, b in rel]
            s = _myfm.GibbsSession(K, 0.1, Xl, rbs, yl, 42, _config(gi), allreduce=ls.callback(rank), n_total_rows=n,
                                   row_offset=lo, main_levels=levels, shard_rank=rank, shard_world=world)
            assert s.plan_flags() & 512, "the cell path was not taken"
            for it in range(n_iter):
                s.step()
            out[rank] = (s.fm.w0, np.asarray(s.fm.w), np.asarray(s.fm.V), s.residual(), lo, float(s.hyper.alpha))
        except BaseException as ex:  # noqa
            errs.append(ex)
            ls.bar.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join(timeout=300)
    assert not errs, errs
    t = oracle.OracleTrainer(main, y, blocks, rank=K, group_index=gi)
    for it in range(n_iter):
        t.step()
    w0, w, V = t.fm()
    e = t.e(main.shape[0])
    for rank in range(world):
        gw0, gw, gV, ge, lo, galpha = out[rank]
        assert abs(gw0 - w0) < 1e-7
        np.testing.assert_allclose(gw, w, rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(gV, V, rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(ge, e[lo:lo + ge.shape[0]], rtol=1e-7, atol=1e-7)
        assert abs(galpha - t.hyper()["alpha"]) < 1e-7 * galpha
        assert np.array_equal(gV, out[0][2]) and np.array_equal(gw, out[0][1])  # replicas bit-identical


def test_cell_path_empty_shard(oracle, monkeypatch):
    # a rank without rows on the cell path: it plans nothing, launches nothing, and still takes part in every agreement of the
    # planner and in every all-reduce of the sweeps (its dense arrays are zero)
    from myfm_amd import _capi, _myfm

    from .test_gpu_cell import SHAPES

    monkeypatch.setenv("MFM_CELL_MIN_ROWS", "0")
    monkeypatch.setenv("MFM_CELL_GROUPS", "5")
    main, blocks, y, shapes = ds.tuple_design(**SHAPES["u_i_ctx"])
    n = main.shape[0]
    gi = ds.group_index_from_shapes(shapes)
    K, world = 2, 3
    cuts = [0, 0, 27000, n]  # rank 0 holds no rows
    ls = Lockstep(world)
    levels = _capi.column_levels(main)[0]
    out, errs = {}, []

    def run(rank):
        try:
            lo, hi = cuts[rank], cuts[rank + 1]
            rbs = [_myfm.RelationBlock(np.asarray(m, dtype=np.int64)[lo:hi], b) for m, b in blocks]
            s = _myfm.GibbsSession(K, 0.1, main[lo:hi], rbs, y[lo:hi], 42, _config(gi), allreduce=ls.callback(rank), n_total_rows=n,
                                   row_offset=lo, main_levels=levels, shard_rank=rank, shard_world=world)
            assert s.plan_flags() & 512, "the cell path was not taken"
            for it in range(3):
                s.step()
            out[rank] = (np.asarray(s.fm.w), np.asarray(s.fm.V), s.residual(), lo)
        except BaseException as ex:  # noqa
            errs.append(ex)
            ls.bar.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join(timeout=300)
    assert not errs, errs
    t = oracle.OracleTrainer(main, y, blocks, rank=K, group_index=gi)
    for it in range(3):
        t.step()
    _, w, V = t.fm()
    for rank in range(world):
        gw, gV, ge, lo = out[rank]
        np.testing.assert_allclose(gw, w, rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(gV, V, rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(ge, t.e(n)[lo:lo + ge.shape[0]], rtol=1e-7, atol=1e-7)
