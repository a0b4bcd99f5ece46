"""world_size-2 `gloo` test (CPU) of the row-sharded decomposition: two processes each hold
half of the rows; per level they all-reduce the partial sufficient statistics of every column and draw
the same coordinates. The result must equal the oracle's unsharded sweep. The statistics/apply arithmetic
is numpy here (FMTrainer.hpp:343-376 per column); what is under test is the partition, the
level-by-level all-reduce protocol and the replicated-draw logic the GPU path uses."""
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sps

from . import datasets as ds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from myfm_amd import _capi
    from myfm_amd.distributed import shard_rows
    from oracle import oracle as O

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    X, y, shapes = ds.onehot_mf(4000, 60, 25, seed=2)
    X = sps.hstack([X, sps.csr_matrix(np.random.default_rng(0).normal(size=(4000, 1)))]).tocsr()  # + a dense column
    gi = np.concatenate([ds.group_index_from_shapes(shapes), [2]]).astype(np.int32)
    K = 3
    t = O.OracleTrainer(X, y, rank=K, group_index=gi)  # replicated start state (same seed everywhere)
    w0, w, V = t.fm()
    e_full = t.e(X.shape[0])
    Xl, yl, _, lo, n = shard_rows(X, y, [], rank, world)
    hi = lo + Xl.shape[0]
    e, q = e_full[lo:hi].copy(), np.zeros(hi - lo)
    level, n_levels = _capi.column_levels(X)  # the schedule is a property of the GLOBAL design
    Xc = Xl.tocsc()
    hyp = t.hyper()
    alpha, lam, mu = hyp["alpha"], hyp["lambda_V"], hyp["mu_V"]
    f = 0
    z = t.clone().rng_sample_normals(X.shape[1])  # replicated variates
    q[:] = Xl.dot(V[:, f])
    v = V[:, f].copy()
    for lv in range(n_levels):
        cols = np.where(level == lv)[0]
        S = np.zeros((len(cols), 2))
        for k, j in enumerate(cols):
            rows, x = Xc.indices[Xc.indptr[j]:Xc.indptr[j + 1]], Xc.data[Xc.indptr[j]:Xc.indptr[j + 1]]
            h = x * (q[rows] - x * v[j])
            S[k] = [(-e[rows] * h).sum(), (h * h).sum()]
        St = torch.from_numpy(S)
        dist.all_reduce(St)  # one collective per level: 2 |level| doubles
        for k, j in enumerate(cols):
            S1, S2 = S[k]
            g = gi[j]
            lin = (S1 + S2 * v[j]) * alpha + lam[g, f] * mu[g, f]
            sq = S2 * alpha + lam[g, f]
            new = lin / sq + z[j] / np.sqrt(sq)
            rows, x = Xc.indices[Xc.indptr[j]:Xc.indptr[j + 1]], Xc.data[Xc.indptr[j]:Xc.indptr[j + 1]]
            h = x * (q[rows] - x * v[j])
            q[rows] += x * (new - v[j])
            e[rows] += h * (new - v[j])
            v[j] = new
    t.update_V_factor(f)
    want_V = t.fm()[2][:, f]
    want_e = t.e(X.shape[0])[lo:hi]
    out.put((rank, float(np.abs(v - want_V).max()), float(np.abs(e - want_e).max()), n_levels))
    dist.destroy_process_group()


def test_row_sharded_sweep_world2_gloo(oracle):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29500 + (os.getpid() % 500)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, dv, de, n_levels in res:
        assert n_levels == 3 and dv < 1e-10 and de < 1e-9, (rank, dv, de)


def test_row_range_partition():
    from myfm_amd.distributed import row_range, shard_rows

    for n, w in [(10, 3), (7, 8), (1000, 4)]:
        parts = [row_range(n, r, w) for r in range(w)]
        assert parts[0][0] == 0 and parts[-1][1] == n
        assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
        assert max(b - a for a, b in parts) - min(b - a for a, b in parts) <= 1
    X = sps.csr_matrix(np.arange(20.0).reshape(10, 2))
    y = np.arange(10.0)
    blocks = [(np.arange(10) % 3, sps.csr_matrix(np.eye(3)))]
    Xl, yl, rel, lo, n = shard_rows(X, y, blocks, 1, 3)
    assert (lo, n) == (4, 10) and Xl.shape[0] == 3 and (yl == [4, 5, 6]).all() and (rel[0][0] == [1, 2, 0]).all()


def _cuts_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from myfm_amd import distributed as D

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    D.enable(set_device=False)
    assert D.active() and D.rank_world() == (rank, world)
    X, y, shapes = ds.onehot_mf(5000, 70, 30, seed=4)
    cuts = D.shard_cuts(X.indices[X.indptr[:-1]], world)
    box = [cuts]
    dist.broadcast_object_list(box, src=0)
    assert box[0] == cuts  # every rank derives the same partition from the same data
    first = X.indices[X.indptr[:-1]]
    for c in cuts[1:-1]:
        assert first[c] != first[c - 1]  # cut between two first-level columns
    assert cuts[0] == 0 and cuts[-1] == 5000 and all(b >= a for a, b in zip(cuts, cuts[1:]))
    assert abs((cuts[rank + 1] - cuts[rank]) - 5000 // world) < 500
    D.disable()
    assert not D.active()
    dist.destroy_process_group()
    out.put((rank, "ok"))


def test_fit_sharding_plan_two_ranks():
    """MyFM*.fit() under myfm_amd.distributed.enable(): the row partition every rank computes (host logic; the device side
    of the same path is tests/test_gpu_sharded.py)."""
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29500 + (os.getpid() % 400) + 17
    ps = [ctx.Process(target=_cuts_worker, args=(r, 2, port, out)) for r in range(2)]
    for p_ in ps:
        p_.start()
    got = sorted(out.get(timeout=240) for _ in ps)
    for p_ in ps:
        p_.join(60)
    assert got == [(0, "ok"), (1, "ok")]


class _FakePeerSession:
    """what myfm_amd.distributed.connect_peers needs of a training context, without a GPU"""

    def __init__(self, rank, pending=True, fail_export=False, fail_import=False):
        self.rank, self.pending, self.fail_export, self.fail_import = rank, pending, fail_export, fail_import
        self.imported, self.dropped = None, False

    def peer_info(self):
        return (self.pending, 1, 2, 64, 1024)

    def peer_export(self):
        if self.fail_export:
            raise RuntimeError("no handle on this rank")
        return bytes([self.rank]) * 256

    def peer_import(self, world, rank, blob):
        if self.fail_import:
            raise RuntimeError("cannot map the peers")
        self.imported = (world, rank, blob)

    def peer_drop(self):
        self.dropped = True


def _peers_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from myfm_amd.distributed import connect_peers

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = []
    # every rank has a layout: the handles arrive in rank order on every rank
    s = _FakePeerSession(rank)
    live = connect_peers(s)
    res.append((live, s.imported == (world, rank, b"".join(bytes([r]) * 256 for r in range(world))), s.dropped))
    # nothing waiting (mfm_finalize's agreement said no): no traffic, nothing installed
    s = _FakePeerSession(rank, pending=False)
    res.append((connect_peers(s), s.imported is None, s.dropped))
    # one rank cannot export / cannot map: EVERY rank gives the path up
    s = _FakePeerSession(rank, fail_export=(rank == 1))
    res.append((connect_peers(s), s.imported is None, s.dropped))
    s = _FakePeerSession(rank, fail_import=(rank == 0))
    res.append((connect_peers(s), True, s.dropped))
    out.put((rank, res))
    dist.destroy_process_group()


def test_connect_peers_protocol_world2_gloo():
    """the hand-over of the row-sharded persistent sweep's exchange buffers (myfm_amd.distributed.connect_peers): all ranks install
    all ranks' handles, or all ranks drop the path -- two gloo processes, no GPU"""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29500 + ((os.getpid() + 137) % 500)
    procs = [ctx.Process(target=_peers_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(out.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in (0, 1):
        ok, none, fe, fi = res[rank]
        assert ok == (True, True, False), (rank, ok)
        assert none == (False, True, False), (rank, none)
        assert fe == (False, True, True), (rank, fe)
        assert fi[0] is False and fi[2] is True, (rank, fi)
