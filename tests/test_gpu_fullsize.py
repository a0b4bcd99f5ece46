"""BASELINE config 3 at FULL size (N = 10 M rows, nnz = 20 M, D = 80 555, rank 32) through the C ABI: the CPU oracle
needs ~6 s per iteration there, so parity is checked through size-independent properties of the sampler instead:

  * the residual the sweeps maintain incrementally (e += h * delta per touched entry, FMTrainer.hpp:374) equals the
    residual update_e recomputes from scratch (:494) -- any lost, doubled or misplaced update of any of the
    2 x 10 M x 33 entry visits of an iteration shows up here;
  * q_train after update_V is X V[:, K-1] (:320, :373);
  * the recomputed residual is the closed-form FM score minus y (FM.hpp:78-135) on a row sample;
  * the chain is bit-reproducible (no order-dependent sums anywhere on the path).
"""
import numpy as np
import pytest

from . import datasets as ds
from .gibbs_driver import CapiGibbs

pytestmark = pytest.mark.gpu

N, NU, NI, K = 10_000_000, 69878, 10677, 32


@pytest.fixture(scope="module")
def capi():
    from myfm_amd import _capi

    if _capi.lib().mfm_device_count() < 1:
        pytest.fail("no HIP device visible: the gpu tests need a real MI355X")
    return _capi


@pytest.fixture(scope="module")
def design():
    X, y, shapes = ds.movielens_like(N, NU, NI, rank_true=32, seed=1)
    return X, y, ds.group_index_from_shapes(shapes)


def _start(capi, oracle, X, y, gi, fused):
    t = oracle.OracleTrainer(X, y, rank=K, group_index=gi)  # initial weights, residual and generator state only
    c = capi.Context(X, y, rank=K, group_index=gi)
    c.set_state(*t.fm())
    c.set_e(t.e(N))
    drv = CapiGibbs(c, None, N, gi, fused=fused)
    drv.use_device_rng(*t.rng_state())
    return c, drv


@pytest.mark.parametrize("fused", [True, False])
def test_full_size_invariants(capi, oracle, design, fused):
    # fused = True is the call bench.py times: update_w0's shift + update_w + update_V as ONE mfm_sweep_wV call with the
    # device's own variates (the persistent launch of mfm_res.hpp); False drives mfm_sweep_w + mfm_sweep_V separately
    import gc

    X, y, gi = design
    c, drv = _start(capi, oracle, X, y, gi, fused)
    flags = c.plan_flags()
    import os

    lazy = os.environ.get("MFM_PLAN_CHECK") is None  # (production mode: the generic plans are built on demand, plan_flags 262)
    assert flags["resident"] and ((flags["soa"] and flags["mf"]) or lazy)  # the path bench.py measures
    # while this context holds the device's CUs for its persistent sweep a second one must not get them (two persistent
    # launches would starve each other's grid barrier): it falls back to the per-factor passes
    c_other = capi.Context(X, y, rank=K, group_index=gi)
    assert not c_other.plan_flags()["resident"] and c_other.plan_flags()["mf"]
    del c_other
    gc.collect()
    seen = {}
    for it in range(2):
        drv.step(before_update_e=lambda: seen.update(e=c.get_e(), q=c.get_q()))
        e_new = c.get_e()
        w0, w, V = c.get_state()
        scale = np.abs(e_new).max()
        # incremental residual == recomputed residual (rounding of ~66 updates per row only)
        assert np.abs(seen["e"] - e_new).max() < 1e-9 * max(scale, 1.0)
        # q-cache of the last factor
        np.testing.assert_allclose(seen["q"], X @ V[:, K - 1], rtol=1e-11, atol=1e-12)
        # closed-form score on a row sample
        rows = np.random.default_rng(it).choice(N, size=200_000, replace=False)
        rows.sort()
        want = ds.fm_score(X[rows], w0, w, V) - y[rows]
        np.testing.assert_allclose(e_new[rows], want, rtol=1e-10, atol=1e-10)
        assert np.isfinite(V).all() and drv.alpha > 0
    v_first = V
    # bit-reproducible (the first context gives its CUs back before the second asks for them)
    del c, drv, seen
    gc.collect()
    c2, drv2 = _start(capi, oracle, X, y, gi, fused)
    assert c2.plan_flags()["resident"]
    for it in range(2):
        drv2.step()
    assert np.array_equal(c2.get_state()[2], v_first)


def test_config5_full_size_n50m_rank64(capi, oracle):
    """BASELINE configs[4] at its OWN size: N = 50 M rows (nnz = 100 M), two one-hot fields + 4 relation blocks, rank 64.
    The CPU oracle needs ~2 minutes per iteration there, so the size-independent properties are checked:
    * regression twin (the same design, the targets taken as real numbers): update_w and factor 0 of update_V against the
      ORACLE at this size (1e-9); after one full update_all the incrementally
      maintained residual == the residual recomputed from scratch (every update of every row by the (2 + 4 blocks) x 65
      sweeps is accounted for), == the closed-form FM score minus y on 50 000 sampled rows;
    * the ordered-probit task itself (5 classes, MyFMOrderedProbit's trainer): two runs of two iterations agree bit for
      bit, the cutpoints are ordered and finite, the Metropolis step ran."""
    import scipy.sparse as sps

    from myfm_amd import _myfm

    K = 64
    main, blocks, y, shapes = ds.config5_like(1.0, ordered=True)
    N = main.shape[0]
    assert N == 50_000_000 and main.nnz == 100_000_000 and len(blocks) == 4
    gi = ds.group_index_from_shapes(shapes)
    # ---- regression twin
    c = capi.Context(main, y, blocks, rank=K, group_index=gi)
    rng = np.random.default_rng(0)
    D = c.D
    c.set_state(0.1, rng.normal(size=D) * 0.1, rng.normal(size=(D, K)) * 0.1)
    c.update_e_regression()
    drv = CapiGibbs(c, None, N, gi)
    t = oracle.OracleTrainer(*ds.toy(), rank=2, seed=7)  # only a seeded mt19937 state to hand to the device
    drv.use_device_rng(*t.rng_state())
    seen = {}
    drv.step(before_update_e=lambda: seen.update(e=c.get_e()))
    e_new = c.get_e()
    assert np.abs(seen["e"] - e_new).max() < 1e-8 * max(1.0, np.abs(e_new).max())
    w0, w, V = c.get_state()
    assert np.isfinite(V).all()
    rows = np.sort(np.random.default_rng(1).choice(N, size=50_000, replace=False))
    X_rows = sps.hstack([main[rows]] + [B[m[rows]] for m, B in blocks]).tocsr()
    np.testing.assert_allclose(e_new[rows], ds.fm_score(X_rows, w0, w, V) - y[rows], rtol=1e-9, atol=1e-9)
    assert c.plan_flags()["cell"]  # update_V on index tuples (mfm_cell.hpp)
    assert c.plan_flags()["streamed_chain"]  # the big blocks' feature sweeps as one pipelined launch each (mfm_chain_stream.hpp)
    del drv, seen, e_new
    # ---- the same twin against the ORACLE at this size: a whole oracle iteration is ~2 minutes, but update_w and the first
    # factor of update_V (FMTrainer.hpp:231-313, :315-482 for f = 0) are seconds after the oracle's ~40 s of setup
    t = oracle.OracleTrainer(main, y, blocks, rank=K, group_index=gi)
    w0, w, V = t.fm()
    c.set_state(w0, w, V)
    c.set_e(t.e(N))
    G = t.G
    hrng = np.random.default_rng(3)
    lam_w, mu_w = hrng.uniform(0.5, 2.0, size=G), hrng.normal(size=G) * 0.1
    lam_V, mu_V = hrng.uniform(0.5, 2.0, size=(G, K)), hrng.normal(size=(G, K)) * 0.1
    t.set_hyper(0.8, mu_w, lam_w, mu_V, lam_V)
    tz = oracle.OracleTrainer(*ds.toy(), rank=2, seed=1)  # replays the variates the big trainer is about to consume
    tz.set_rng_state(*t.rng_state())
    zw = tz.rng_sample_normals(D)
    zv = tz.rng_sample_normals(D)
    t.substep(4)           # update_w
    t.update_V_factor(0)   # update_V, factor 0
    c.sweep_w(0.8, lam_w, mu_w, zw)
    c.sweep_V(0, 1, 0.8, lam_V, mu_V, zv)
    _, tw, tV = t.fm()
    _, gw, gV = c.get_state()
    np.testing.assert_allclose(gw, tw, rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(gV[:, 0], tV[:, 0], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(c.get_e(), t.e(N), rtol=1e-8, atol=1e-8)
    del c, t, tz
    # ---- the ordered-probit task, twice
    rels = [_myfm.RelationBlock(np.asarray(m, dtype=np.int64), B) for m, B in blocks]
    b = _myfm.ConfigBuilder()
    b.set_alpha_0(1.0).set_beta_0(1.0).set_gamma_0(1.0).set_mu_0(0.0).set_reg_0(1.0)
    b.set_group_index([int(g) for g in gi]).set_n_iter(2).set_n_kept_samples(1)
    b.set_task_type(_myfm.TaskType.ORDERED)
    b.set_cutpoint_groups([(5, np.arange(N))])
    cfg = b.build()
    runs = []
    for rep in range(2):
        predictor, history = _myfm.create_train_fm(K, 0.1, main, rels, y, 42, cfg, lambda *a: False)
        fm = predictor.samples[-1]
        runs.append((np.array(fm.V), np.array(fm.w), fm.w0, np.array(fm.cutpoints[0])))
        assert len(history.hypers) == 2
        del predictor
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1]) and runs[0][2] == runs[1][2]
    assert np.array_equal(runs[0][3], runs[1][3])
    cp = runs[0][3]
    assert cp.shape == (4,) and np.isfinite(cp).all() and np.all(np.diff(cp) > 0)
    assert np.isfinite(runs[0][0]).all()


def test_config3_full_size_two_ranks_persistent_sweep(oracle, design, monkeypatch):
    """BASELINE configs[2] at full size (N = 10 M, rank 32) row-sharded over TWO ranks that share this GPU (124 CUs each, sessions of
    one process): the persistent sweep on both, item sums and first-level coefficients exchanged inside the launches (mfm_res.hpp
    XCH), one collective per iteration. Two full iterations of the trainer against the CPU oracle's unsharded chain (the
    oracle takes ~6 s per iteration here); replicas bit-identical."""
    import threading

    from myfm_amd import _capi, _myfm
    from myfm_amd.distributed import shard_cuts

    from .test_gpu_sharded import Lockstep, _config

    monkeypatch.setenv("MFM_RES_CUS", "124")
    X, y, gi = design
    world, iters = 2, 2
    cuts = shard_cuts(X.indices[X.indptr[:-1]], world)
    ls = Lockstep(world)
    levels = _capi.column_levels(X)[0]
    out, errs, peers = {}, [], {}
    meet = threading.Barrier(world)

    def run(rank):
        try:
            lo, hi = cuts[rank], cuts[rank + 1]
            s = _myfm.GibbsSession(K, 0.1, X[lo:hi], [], y[lo:hi], 42, _config(gi), allreduce=ls.callback(rank), n_total_rows=N,
                                   row_offset=lo, main_levels=levels, shard_rank=rank, shard_world=world)
            info = s.peer_info()
            assert info[0], "no layout waiting for the peers"
            peers[rank] = (info[1], info[2]) + tuple(s.peer_model_info())
            meet.wait()
            s.peer_set(world, rank, [peers[r][0] for r in range(world)], [peers[r][1] for r in range(world)])
            s.peer_set_model(world, rank, [peers[r][2] for r in range(world)], [peers[r][3] for r in range(world)])
            calls0 = ls.counts[rank]
            for it in range(iters):
                s.step()
            out[rank] = (s.plan_flags(), s.fm.w0, np.asarray(s.fm.w), np.asarray(s.fm.V), float(s.hyper.alpha), ls.counts[rank] - calls0)
        except BaseException as ex:  # noqa
            errs.append(ex)
            ls.bar.abort()
            meet.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join(timeout=900)
    assert not errs, errs
    t = oracle.OracleTrainer(X, y, rank=K, group_index=gi)
    for it in range(iters):
        t.step()
    w0, w, V = t.fm()
    for rank in range(world):
        flags, gw0, gw, gV, alpha, calls = out[rank]
        assert flags & 256 and flags & 8 and calls <= 2 * iters, (flags, calls)
        assert abs(gw0 - w0) < 1e-7 and abs(alpha - t.hyper()["alpha"]) < 1e-7 * alpha
        np.testing.assert_allclose(gw, w, rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(gV, V, rtol=1e-7, atol=1e-8)
        assert np.array_equal(gV, out[0][3]) and np.array_equal(gw, out[0][2])


def test_config3_full_size_one_rank_vs_oracle(oracle, design):
    """The exact call bench.py times -- `GibbsSession` on ONE rank (device variates, residual policy 1: the launch does not
    write the residual back, `update_e` recomputes it; bench.py's own `make_config`) -- at BASELINE configs[2]'s full size
    (N = 10 M, rank 32), two `update_all` iterations (BaseFMTrainer.hpp:135-152) against the CPU oracle's chain. The oracle
    needs ~6 s per iteration here. `plan_flags` must be the ones the bench line reports for a one-GPU run."""
    import gc

    import bench
    from myfm_amd import _myfm

    X, y, gi = design
    cfg = bench.make_config(_myfm, gi, 10, 0, "regression", N)
    sess = _myfm.GibbsSession(K, 0.1, X, [], y, 42, cfg)
    flags = int(sess.plan_flags())
    assert flags & 256, flags  # the persistent sweep (mfm_res.hpp) -- what bench.py's config 3 line runs on one GPU
    import os

    if os.environ.get("MFM_PLAN_CHECK") is None:
        assert flags == 262, flags  # production mode: generic plans on demand -- the bench line's `plan_flags`
    t = oracle.OracleTrainer(X, y, rank=K, group_index=gi)
    for it in range(2):
        sess.step()
        t.step()
        sess.synchronize()
        w0, w, V = t.fm()
        hy = t.hyper()
        assert abs(sess.fm.w0 - w0) < 1e-7 and abs(sess.hyper.alpha - hy["alpha"]) < 1e-7 * hy["alpha"]
        np.testing.assert_allclose(np.asarray(sess.fm.w), w, rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(np.asarray(sess.fm.V), V, rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(np.asarray(sess.hyper.lambda_V), hy["lambda_V"], rtol=1e-7)
        np.testing.assert_allclose(np.asarray(sess.hyper.mu_V), hy["mu_V"], rtol=1e-7, atol=1e-8)
    # the residual after update_e, in row order, against the oracle's
    np.testing.assert_allclose(np.asarray(sess.residual()), t.e(N), rtol=1e-7, atol=1e-7)
    del sess
    gc.collect()


def test_config5_own_task_scale01_vs_oracle(oracle):
    """BASELINE configs[4]'s OWN task -- ordered probit, rank 64, two one-hot fields + 4 relation blocks -- at scale 0.1
    (N = 5 M rows, nnz = 10 M) draw for draw against the CPU oracle: `exact_latent_draws` keeps the trainer's std::mt19937 on
    the host, so that the latent draws (OProbitSampler.hpp:238-272), the cutpoint Metropolis step (:359-387) and every
    hyper-parameter / w / V variate are the reference's own, while every O(N) and O(nnz) step runs on the device. Two
    iterations (the oracle needs ~11 s each): kept samples, cutpoints, hyper-parameter draws and the Metropolis accept count."""
    from myfm_amd import _myfm

    from .test_gpu_baseline_configs import _assert_chain

    Kc, n_iter = 64, 2
    main, blocks, y, shapes = ds.config5_like(0.1, ordered=True)
    n = main.shape[0]
    assert n == 5_000_000 and len(blocks) == 4
    gi = ds.group_index_from_shapes(shapes)
    rels = [_myfm.RelationBlock(np.asarray(m, dtype=np.int64), B) for m, B in blocks]
    b = _myfm.ConfigBuilder()
    b.set_alpha_0(1.0).set_beta_0(1.0).set_gamma_0(1.0).set_mu_0(0.0).set_reg_0(1.0)
    b.set_group_index([int(g) for g in gi]).set_n_iter(n_iter).set_n_kept_samples(n_iter)
    b.set_task_type(_myfm.TaskType.ORDERED)
    b.set_cutpoint_groups([(5, np.arange(n))])
    b.set_exact_latent_draws(True)
    predictor, history = _myfm.create_train_fm(Kc, 0.1, main, rels, y, 42, b.build(), lambda *a: False)
    t = oracle.OracleTrainer(main, y, blocks, rank=Kc, group_index=gi, task=oracle.ORDERED, cutpoint_groups=[(5, np.arange(n))])
    samples, hypers, cuts = [], [], []
    for it in range(n_iter):
        t.step()
        samples.append(t.fm())
        hypers.append(t.hyper())
        cuts.append(t.cutpoints(0))
    _assert_chain(predictor, history, samples, hypers)
    for fm, cut in zip(predictor.samples, cuts):
        np.testing.assert_allclose(fm.cutpoints[0], cut, rtol=1e-7, atol=1e-7)
    assert list(history.n_mh_accept) == [t.mh_accept(0)]


def test_beyond_on_chip_capacity_full_size_invariants(capi, oracle):
    """A table LARGER than the persistent sweep's on-chip capacity (10.48 M rows per GPU): BASELINE configs[2]'s users / items with
    N = 14 M rows. The sweep keeps 80 slots per thread on chip and streams the residual of the others every sweep (`resident_overflow`).
    Size-independent properties over two full iterations of the call bench.py times: the incrementally maintained residual == the
    one recomputed from scratch, == the closed-form FM score - y on a row sample; finite state; a second context bit for bit."""
    import gc

    n = 14_000_000
    X, y, shapes = ds.movielens_like(n, NU, NI, rank_true=32, seed=1)
    gi = ds.group_index_from_shapes(shapes)

    def start():
        t = oracle.OracleTrainer(X[:1000], y[:1000], rank=2, seed=3)  # (only a seeded mt19937 state to hand to the device)
        c = capi.Context(X, y, rank=K, group_index=gi)
        rng = np.random.default_rng(0)
        c.set_state(0.1, rng.normal(size=c.D) * 0.1, rng.normal(size=(c.D, K)) * 0.1)
        c.update_e_regression()
        drv = CapiGibbs(c, None, n, gi, fused=True)
        drv.use_device_rng(*t.rng_state())
        return c, drv

    c, drv = start()
    flags = c.plan_flags()
    assert flags["resident"] and flags["resident_overflow"], flags
    seen = {}
    for it in range(2):
        drv.step(before_update_e=lambda: seen.update(e=c.get_e()))
        e_new = c.get_e()
        w0, w, V = c.get_state()
        assert np.abs(seen["e"] - e_new).max() < 1e-9 * max(np.abs(e_new).max(), 1.0)
        rows = np.sort(np.random.default_rng(it).choice(n, size=100_000, replace=False))
        np.testing.assert_allclose(e_new[rows], ds.fm_score(X[rows], w0, w, V) - y[rows], rtol=1e-10, atol=1e-10)
        assert np.isfinite(V).all() and drv.alpha > 0
    v_first = V
    del c, drv, seen
    gc.collect()
    c2, drv2 = start()
    for it in range(2):
        drv2.step()
    assert np.array_equal(c2.get_state()[2], v_first)
