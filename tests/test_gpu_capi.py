"""GPU parity tests through the C ABI (libmyfm_hip.so) against the CPU oracle.

Tolerances: the device sums each column's sufficient statistics with a fixed tree instead of the
reference's sequential order, so single draws agree to ~1e-12 relative; the state after a few
iterations to 1e-8 (SURVEY 8d asks <= 1e-9 after 1 and <= 1e-6 after 10 iterations).
"""
import numpy as np
import pytest
import scipy.sparse as sps

from . import datasets as ds
from .gibbs_driver import CapiGibbs

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _small_tables_take_the_persistent_sweep(monkeypatch):
    """The persistent sweep is the default from 2^20 rows on; these tests mean to cover it on small tables whatever mode the suite
    runs in (conftest sets the same for the checker mode; MYFM_TEST_PRODUCTION=1 does not)."""
    monkeypatch.setenv("MFM_RES_MIN_ROWS", "0")


@pytest.fixture(scope="module")
def capi():
    from myfm_amd import _capi

    if _capi.lib().mfm_device_count() < 1:
        pytest.fail("no HIP device visible: the gpu tests need a real MI355X")
    return _capi


def _designs():
    out = {}
    X, y = ds.toy()
    out["toy_dense_plus_onehot"] = (X, y, None, 4)
    X, score = ds.middle_data(1000)
    out["middle_multilevel"] = (X, score + np.random.RandomState(0).normal(size=score.shape), None, 3)
    X, y, shapes = ds.onehot_mf(20000, 300, 40, seed=1)  # items with > 4096 ratings: long-column path
    out["onehot_long_columns"] = (X, y, ds.group_index_from_shapes(shapes), 8)
    X, y, shapes = ds.onehot_mf(30000, 2000, 700, seed=2, sort_by_user=False)
    out["onehot_unsorted"] = (X, y, ds.group_index_from_shapes(shapes), 5)
    return out


DESIGNS = _designs()


def _checked():
    import os

    return os.environ.get("MFM_PLAN_CHECK") is not None  # (tests/conftest.py: the default; MYFM_TEST_NO_PLAN_CHECK=1 turns it off)


def _pair(oracle, capi, X, y, gi, rank, blocks=(), **kw):
    t = oracle.OracleTrainer(X, y, blocks, rank=rank, group_index=gi, **kw)
    n = X.shape[0]
    D = t.D
    if gi is None:
        gi = np.zeros(D, dtype=np.int32)
    c = capi.Context(X, y, blocks, rank=rank, group_index=gi)
    c.set_state(*t.fm())
    c.set_e(t.e(n))
    return t, c, gi


@pytest.mark.parametrize("name", list(DESIGNS))
def test_sweep_V_single_factor(oracle, capi, name):
    X, y, gi, rank = DESIGNS[name]
    t, c, gi = _pair(oracle, capi, X, y, gi, rank)
    n, D, G = X.shape[0], t.D, t.G
    rng = np.random.default_rng(5)
    lam = rng.uniform(0.5, 2.0, size=(G, rank))
    mu = rng.normal(size=(G, rank)) * 0.1
    h = t.hyper()
    t.set_hyper(0.7, h["mu_w"], h["lambda_w"], mu, lam)
    for f in range(min(rank, 2)):
        z = t.clone().rng_sample_normals(D)
        t.update_V_factor(f)
        c.sweep_V(f, f + 1, 0.7, lam, mu, z)
        w0, w, V = t.fm()
        gw0, gw, gV = c.get_state()
        np.testing.assert_allclose(gV[:, f], V[:, f], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(c.get_e(), t.e(n), rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(c.get_q(), t.q(n), rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("name", list(DESIGNS))
def test_sweep_w(oracle, capi, name):
    X, y, gi, rank = DESIGNS[name]
    t, c, gi = _pair(oracle, capi, X, y, gi, rank)
    n, D, G = X.shape[0], t.D, t.G
    rng = np.random.default_rng(6)
    lam, mu = rng.uniform(0.5, 2.0, size=G), rng.normal(size=G) * 0.1
    h = t.hyper()
    t.set_hyper(1.3, mu, lam, h["mu_V"], h["lambda_V"])
    z = t.clone().rng_sample_normals(D)
    t.substep(4)
    c.sweep_w(1.3, lam, mu, z)
    np.testing.assert_allclose(c.get_state()[1], t.fm()[1], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(c.get_e(), t.e(n), rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("name", list(DESIGNS))
def test_update_e_and_reductions(oracle, capi, name):
    X, y, gi, rank = DESIGNS[name]
    t, c, gi = _pair(oracle, capi, X, y, gi, rank)
    n = X.shape[0]
    t.substep(8)
    c.update_e_regression()
    e = t.e(n)
    np.testing.assert_allclose(c.get_e(), e, rtol=1e-11, atol=1e-11)
    se, se2 = c.reduce_e()
    np.testing.assert_allclose([se, se2], [e.sum(), (e * e).sum()], rtol=1e-11)
    c.shift_e(0.25)
    np.testing.assert_allclose(c.get_e(), e + 0.25, rtol=1e-14)
    w0, w, V = t.fm()
    G = t.G
    mu_w = np.linspace(-0.1, 0.1, G)
    s, ssd = c.group_stats_w(mu_w)
    for g in range(G):
        sel = gi == g
        np.testing.assert_allclose(s[g], w[sel].sum(), rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(ssd[g], ((w[sel] - mu_w[g]) ** 2).sum(), rtol=1e-11)
    mu_V = np.random.default_rng(0).normal(size=(G, rank)) * 0.05
    s, ssd = c.group_stats_V(mu_V)
    for g in range(G):
        sel = gi == g
        np.testing.assert_allclose(s[g], V[sel].sum(axis=0), rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(ssd[g], ((V[sel] - mu_V[g]) ** 2).sum(axis=0), rtol=1e-11)


@pytest.mark.parametrize("name", list(DESIGNS))
def test_full_iterations_match_oracle(oracle, capi, name):
    X, y, gi, rank = DESIGNS[name]
    t, c, gi = _pair(oracle, capi, X, y, gi, rank)
    drv = CapiGibbs(c, t.clone(), X.shape[0], gi)
    for it in range(5):
        t.step()
        drv.step()
        tol = 1e-9 if it == 0 else 1e-7
        w0, w, V = t.fm()
        gw0, gw, gV = c.get_state()
        assert abs(drv.w0 - w0) <= tol * max(1, abs(w0))
        np.testing.assert_allclose(gw, w, rtol=tol, atol=tol)
        np.testing.assert_allclose(gV, V, rtol=tol, atol=tol)
        h, gh = t.hyper(), drv.hyper()
        for k in h:
            np.testing.assert_allclose(gh[k], h[k], rtol=tol, atol=tol)
    np.testing.assert_allclose(c.get_e(), t.e(X.shape[0]), rtol=1e-7, atol=1e-7)


@pytest.mark.parametrize("qfree", [True, False])
def test_scattered_level_path(oracle, capi, monkeypatch, qfree):
    # the row-blocked two-pass path (k_scat_*) is chosen for large scattered levels only; force it on a
    # small unsorted design (both one-hot fields jump between far-apart rows) and on non-unit values.
    # qfree: short-row tables recompute q_train on the fly (PMainVe) instead of keeping the q-cache
    monkeypatch.setenv("MFM_SCATTER_MIN_NNZ", "1000")
    monkeypatch.setenv("MFM_TILE_BITS", "0")  # the L2-window kernels (k_scat_*); the row-tile path has its own test
    if qfree:
        monkeypatch.setenv("MFM_QFREE", "1")
    X, y, shapes = ds.onehot_mf(200000, 700, 90, seed=5, sort_by_user=False)
    gi = ds.group_index_from_shapes(shapes)
    for scale in (None, 0.5):
        Xs = X.copy()
        if scale:
            Xs.data = np.where(np.arange(Xs.nnz) % 3 == 0, scale, 1.5)
        t, c, _ = _pair(oracle, capi, Xs, y, gi, 3)
        assert c.plan_flags()["qfree"] == qfree
        drv = CapiGibbs(c, t.clone(), X.shape[0], gi)
        for it in range(3):
            t.step()
            drv.step()
        np.testing.assert_allclose(c.get_q(), t.q(X.shape[0]), rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(c.get_state()[2], t.fm()[2], rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(c.get_state()[1], t.fm()[1], rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(c.get_e(), t.e(X.shape[0]), rtol=1e-7, atol=1e-7)
        assert "sweep_V_scattered" in _timing_classes(c, drv)  # (runs one more sweep: keep it last)


@pytest.mark.parametrize("tile_bits", [9, 12, 13])
def test_row_tile_level_path(oracle, capi, monkeypatch, tile_bits):
    # scattered levels through LDS-staged row tiles (k_tile_*): packed entries, padded tiles, a last partial
    # tile (N not a multiple of the tile), unit and non-unit values, both sweeps (w and V)
    monkeypatch.setenv("MFM_SCATTER_MIN_NNZ", "1000")
    monkeypatch.setenv("MFM_TILE_BITS", str(tile_bits))
    X, y, shapes = ds.onehot_mf(200003, 700, 90, seed=6, sort_by_user=False)
    gi = ds.group_index_from_shapes(shapes)
    for scale in (None, 0.5):
        Xs = X.copy()
        if scale:
            Xs.data = np.where(np.arange(Xs.nnz) % 3 == 0, scale, 1.5)
        t, c, _ = _pair(oracle, capi, Xs, y, gi, 3)
        drv = CapiGibbs(c, t.clone(), X.shape[0], gi)
        for it in range(3):
            t.step()
            drv.step()
        np.testing.assert_allclose(c.get_q(), t.q(X.shape[0]), rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(c.get_state()[2], t.fm()[2], rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(c.get_state()[1], t.fm()[1], rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(c.get_e(), t.e(X.shape[0]), rtol=1e-7, atol=1e-7)
        assert "sweep_V_scattered" in _timing_classes(c, drv)


@pytest.mark.parametrize("n_fields,fused", [(2, True), (2, "mf_k4"), (2, "tile_fused"), (3, True), (2, False), (3, "csr_q"),
                                            (3, "no_multi")])
def test_split_layout_latent_sweep(oracle, capi, monkeypatch, n_fields, fused):
    # update_V with e and q as separate arrays (run_plan_soa): first level (user-sorted, contiguous columns)
    # rebuilds q, middle levels read and write both, the last level writes only e; q_train is restored after
    # fused: the last level's apply pass also runs the next factor's first level on the LDS tile
    # (k_tile_apply_next; tiles aligned to the users' row ranges, a few users longer than a tile)
    # three fields: apply + next-level statistics in one pass, wrap-around pass with the next q from the entry streams
    # ("csr_q": from the CSR rows instead; "no_multi": the per-level passes)
    monkeypatch.setenv("MFM_SCATTER_MIN_NNZ", "1000")
    if not fused:
        monkeypatch.setenv("MFM_NO_FUSED_NEXT", "1")
    if fused == "csr_q":
        monkeypatch.setenv("MFM_NO_FUSED_MULTIQ", "1")
    if fused == "no_multi":
        monkeypatch.setenv("MFM_NO_FUSED_MULTI", "1")
    if fused == "tile_fused":  # two fields without the two-field pass: k_tile_apply_next with the q-cache in HBM
        monkeypatch.setenv("MFM_NO_MF", "1")
    if fused == "mf_k4":  # the two-field pass with 4 rows per thread
        monkeypatch.setenv("MFM_MF_K", "4")
    want_mf = n_fields == 2 and fused in (True, "mf_k4")
    fused = bool(fused)
    import scipy.sparse as sps
    n = 150001
    X, y, shapes = ds.onehot_mf(n, 200, 120, seed=21, sort_by_user=True)
    assert np.diff(X.tocsc().indptr)[:200].max() > 4096
    if n_fields == 3:
        rng = np.random.default_rng(3)
        ctx = rng.integers(0, 37, size=n)
        extra = sps.csr_matrix((np.ones(n), ctx, np.arange(n + 1)), shape=(n, 37))
        X = sps.hstack([X, extra]).tocsr()
        X.sort_indices()
        shapes = shapes + [37]
    gi = ds.group_index_from_shapes(shapes)
    for scale in (None, 0.5):
        Xs = X.copy()
        if scale:
            Xs.data = np.where(np.arange(Xs.nnz) % 3 == 0, scale, 1.5)
        t, c, _ = _pair(oracle, capi, Xs, y, gi, 3)
        f_ = c.plan_flags()
        # (production mode -- no MFM_PLAN_CHECK: a two-field table takes the persistent sweep before the generic plans exist)
        assert (f_["soa"] and f_["fused_next"] == fused and f_["mf"] == want_mf) or (f_["resident"] and not _checked())
        drv = CapiGibbs(c, t.clone(), n, gi)
        for it in range(3):
            t.step()
            drv.step()
        np.testing.assert_allclose(c.get_q(), t.q(n), rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(c.get_state()[2], t.fm()[2], rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(c.get_state()[1], t.fm()[1], rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(c.get_e(), t.e(n), rtol=1e-7, atol=1e-7)


@pytest.mark.parametrize("cus", [None, 12, 4, 1])
def test_resident_latent_sweep(oracle, capi, monkeypatch, cus):
    # update_V of a two-field one-hot table as ONE persistent launch (mfm_res.hpp): a workgroup per CU keeps the residual of
    # its users' rows in registers / LDS for all factors, item statistics leave the CU once per (workgroup, item), grid
    # barriers around the item draw. cus = CUs the planner may use: all (8 slots per thread, ~37 workgroups), 12 (32 slots per
    # thread), 4 (64 register + 16 LDS slots per thread), 1 (a single workgroup: no second arrival at the barriers).
    # Against the oracle draw for draw; a second context reproduces the chain bit for bit.
    monkeypatch.setenv("MFM_SCATTER_MIN_NNZ", "1000")
    if cus:
        monkeypatch.setenv("MFM_RES_CUS", str(cus))
    n = 150001 if cus != 1 else 30011
    X, y, shapes = ds.onehot_mf(n, 200 if cus != 1 else 90, 120, seed=21, sort_by_user=True)
    gi = ds.group_index_from_shapes(shapes)
    chains = []
    for rep in range(2):
        t, c, _ = _pair(oracle, capi, X, y, gi, 3)
        assert c.plan_flags()["resident"] and (c.plan_flags()["mf"] or not _checked())
        drv = CapiGibbs(c, t.clone(), n, gi)
        for it in range(3):
            t.step()
            drv.step()
            np.testing.assert_allclose(c.get_state()[2], t.fm()[2], rtol=1e-7, atol=1e-8, err_msg="iteration %d" % it)
        np.testing.assert_allclose(c.get_q(), t.q(n), rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(c.get_state()[1], t.fm()[1], rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(c.get_e(), t.e(n), rtol=1e-7, atol=1e-7)
        chains.append((c.get_state()[2].copy(), c.get_e().copy()))
        assert "sweep_V_resident" in _timing_classes(c, drv)
    assert np.array_equal(chains[0][0], chains[1][0]) and np.array_equal(chains[0][1], chains[1][1])
    # the per-factor passes (MFM_NO_RESIDENT) walk the same chain
    monkeypatch.setenv("MFM_NO_RESIDENT", "1")
    t, c, _ = _pair(oracle, capi, X, y, gi, 3)
    assert not c.plan_flags()["resident"] and c.plan_flags()["mf"]
    drv = CapiGibbs(c, t.clone(), n, gi)
    for it in range(3):
        drv.step()
    np.testing.assert_allclose(c.get_state()[2], chains[0][0], rtol=1e-9, atol=1e-10)


def test_resident_layout_from_the_device_plans_on_demand(oracle, capi, monkeypatch):
    # f4: the persistent sweep's slot layout is built on the device from the CSR (mfm_res_plan.hpp) BEFORE anything else is
    # planned; when it takes the table, X_t / level plans / row tiles are built only when a call needs them. (Every other
    # resident test runs with MFM_PLAN_CHECK=1: both builders, layouts compared array for array, nothing lazy.)
    monkeypatch.delenv("MFM_PLAN_CHECK", raising=False)
    monkeypatch.setenv("MFM_SCATTER_MIN_NNZ", "1000")
    n = 120001
    X, y, shapes = ds.onehot_mf(n, 260, 140, seed=9, sort_by_user=True)
    gi = ds.group_index_from_shapes(shapes)
    t, c, _ = _pair(oracle, capi, X, y, gi, 3)
    f = c.plan_flags()
    assert f["resident"] and not f["soa"] and not f["mf"]  # nothing but the resident layout exists yet
    drv = CapiGibbs(c, t.clone(), n, gi, fused=True)
    for it in range(2):  # update_w0 + update_w + update_V in one launch, update_e in slot order: no generic structure needed
        t.step()
        drv.step()
        np.testing.assert_allclose(c.get_state()[2], t.fm()[2], rtol=1e-7, atol=1e-8, err_msg="iteration %d" % it)
    assert not c.plan_flags()["soa"]
    np.testing.assert_allclose(c.get_e(), t.e(n), rtol=1e-7, atol=1e-7)
    np.testing.assert_allclose(c.get_q(), t.q(n), rtol=1e-7, atol=1e-8)
    # a stand-alone mfm_sweep_w needs the level plan of the table: built now, the chain goes on
    drv.fused = False
    for it in range(2):
        t.step()
        drv.step()
        np.testing.assert_allclose(c.get_state()[1], t.fm()[1], rtol=1e-7, atol=1e-8, err_msg="iteration %d" % it)
        np.testing.assert_allclose(c.get_state()[2], t.fm()[2], rtol=1e-7, atol=1e-8, err_msg="iteration %d" % it)
    f = c.plan_flags()
    assert f["resident"] and f["soa"] and f["mf"]
    np.testing.assert_allclose(c.get_e(), t.e(n), rtol=1e-7, atol=1e-7)


def test_resident_sweep_leaves_no_residual_when_update_e_follows(oracle, capi, monkeypatch):
    # regression (mfm_set_residual_policy(1)): update_e recomputes e = score - y after every update_V (FMTrainer.hpp:494), so the
    # persistent launch does not write its on-chip residual back; whoever reads the residual between the sweep and update_e gets
    # it recomputed. The chain is the oracle's; the residual read right after the sweep is the oracle's incrementally updated
    # one up to rounding; a second sweep without update_e in between starts from the same residual.
    monkeypatch.setenv("MFM_SCATTER_MIN_NNZ", "1000")
    n = 120001
    X, y, shapes = ds.onehot_mf(n, 260, 140, seed=5, sort_by_user=True)
    gi = ds.group_index_from_shapes(shapes)
    t, c, _ = _pair(oracle, capi, X, y, gi, 4)
    assert c.plan_flags()["resident"]
    c.set_residual_policy(True)
    drv = CapiGibbs(c, t.clone(), n, gi, fused=True)
    seen = {}
    for it in range(3):
        t.step()
        drv.step(before_update_e=lambda: seen.update(e=c.get_e()))
        np.testing.assert_allclose(c.get_state()[2], t.fm()[2], rtol=1e-7, atol=1e-8, err_msg="iteration %d" % it)
        np.testing.assert_allclose(c.get_state()[1], t.fm()[1], rtol=1e-7, atol=1e-8)
        # between the sweep and update_e: recomputed == what update_e is about to write
        np.testing.assert_allclose(seen["e"], c.get_e(), rtol=0, atol=1e-9)
    np.testing.assert_allclose(c.get_e(), t.e(n), rtol=1e-7, atol=1e-7)
    assert "sweep_V_resident" in _timing_classes(c, drv)


@pytest.mark.parametrize("device_rng", [False, True])
def test_resident_linear_and_latent_sweeps_in_one_launch(oracle, capi, monkeypatch, device_rng):
    # mfm_sweep_wV: update_w0's residual shift, update_w and update_V of a two-field one-hot table as ONE persistent launch
    # (the linear sweep is the latent sweep with h = 1), against the oracle's update_all draw for draw; variates from the
    # host (the oracle's generator) and from the device stream
    monkeypatch.setenv("MFM_SCATTER_MIN_NNZ", "1000")
    n = 120001
    X, y, shapes = ds.onehot_mf(n, 260, 140, seed=5, sort_by_user=True)
    gi = ds.group_index_from_shapes(shapes)
    t, c, _ = _pair(oracle, capi, X, y, gi, 4)
    assert c.plan_flags()["resident"]
    drv = CapiGibbs(c, t.clone(), n, gi, fused=True)
    if device_rng:
        state, pos = t.rng_state()
        drv.use_device_rng(state, pos)
    for it in range(4):
        t.step()
        drv.step()
        w0, w, V = t.fm()
        gw0, gw, gV = c.get_state()
        assert abs(gw0 - w0) < 1e-9
        np.testing.assert_allclose(gw, w, rtol=1e-7, atol=1e-9, err_msg="w, iteration %d" % it)
        np.testing.assert_allclose(gV, V, rtol=1e-7, atol=1e-9, err_msg="V, iteration %d" % it)
    np.testing.assert_allclose(c.get_e(), t.e(n), rtol=1e-7, atol=1e-7)
    c.timing_enable(True)
    c.timing_reset()
    drv.step()
    classes = set(c.timing())
    c.timing_enable(False)
    assert "sweep_V_resident" in classes and not any(k.startswith("sweep_w") for k in classes), classes


def test_resident_factor_subranges_and_empty_columns(oracle, capi, monkeypatch):
    # mfm_sweep_V over [0, 2), [2, 3), [3, 5) through the resident launch, on a table with features that never occur in either
    # field (drawn from the prior by the workgroup / item slice they are dealt to)
    monkeypatch.setenv("MFM_SCATTER_MIN_NNZ", "1000")
    n, K = 60001, 5
    X, y, shapes = ds.onehot_mf(n, 300, 90, seed=31, sort_by_user=True)
    import scipy.sparse as sps

    C = X.tocoo()
    keep_u = np.setdiff1d(np.arange(300), [0, 17, 299])  # users / items without rows
    keep_i = np.setdiff1d(np.arange(90), [3, 89])
    u = X.indices[0::2]
    i = X.indices[1::2] - 300
    u2 = keep_u[u % len(keep_u)]
    order = np.argsort(u2, kind="stable")
    u2, i2 = u2[order], keep_i[i[order] % len(keep_i)]
    ind = np.empty(2 * n, dtype=np.int32)
    ind[0::2] = u2
    ind[1::2] = 300 + i2
    X = sps.csr_matrix((np.ones(2 * n), ind, np.arange(0, 2 * n + 1, 2)), shape=(n, 390))
    y = y[order]
    gi = ds.group_index_from_shapes(shapes)
    t, c, _ = _pair(oracle, capi, X, y, gi, K)
    assert c.plan_flags()["resident"]
    G, D = t.G, t.D
    rng = np.random.default_rng(8)
    lam = rng.uniform(0.5, 2.0, size=(G, K))
    mu = rng.normal(size=(G, K)) * 0.1
    h = t.hyper()
    t.set_hyper(0.9, h["mu_w"], h["lambda_w"], mu, lam)
    for f0, f1 in ((0, 2), (2, 3), (3, 5)):
        z = t.clone().rng_sample_normals(D * (f1 - f0)).reshape(f1 - f0, D)
        for f in range(f0, f1):
            t.update_V_factor(f)
        c.sweep_V(f0, f1, 0.9, lam, mu, z)
        np.testing.assert_allclose(c.get_state()[2][:, f0:f1], t.fm()[2][:, f0:f1], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(c.get_e(), t.e(n), rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(c.get_q(), t.q(n), rtol=1e-8, atol=1e-9)


@pytest.mark.parametrize("n_fields", [2, 3])
def test_split_layout_factor_subranges(oracle, capi, monkeypatch, n_fields):
    # mfm_sweep_V over [0, 2), [2, 3), [3, 5): every call starts with an unfused first level, ends with an unfused
    # apply pass and picks its variates / hyper columns by absolute factor index
    monkeypatch.setenv("MFM_SCATTER_MIN_NNZ", "1000")
    n, K = 60001, 5
    X, y, shapes = ds.onehot_mf(n, 300, 90, seed=31, sort_by_user=True)
    if n_fields == 3:
        import scipy.sparse as sps

        ctx = np.random.default_rng(4).integers(0, 23, size=n)
        X = sps.hstack([X, sps.csr_matrix((np.ones(n), ctx, np.arange(n + 1)), shape=(n, 23))]).tocsr()
        X.sort_indices()
        shapes = shapes + [23]
    gi = ds.group_index_from_shapes(shapes)
    t, c, _ = _pair(oracle, capi, X, y, gi, K)
    assert c.plan_flags()["fused_next"] or (c.plan_flags()["resident"] and not _checked())
    G, D = t.G, t.D
    rng = np.random.default_rng(8)
    lam = rng.uniform(0.5, 2.0, size=(G, K))
    mu = rng.normal(size=(G, K)) * 0.1
    h = t.hyper()
    t.set_hyper(0.9, h["mu_w"], h["lambda_w"], mu, lam)
    for f0, f1 in ((0, 2), (2, 3), (3, 5)):
        z = t.clone().rng_sample_normals(D * (f1 - f0)).reshape(f1 - f0, D)
        for f in range(f0, f1):
            t.update_V_factor(f)
        c.sweep_V(f0, f1, 0.9, lam, mu, z)
        np.testing.assert_allclose(c.get_state()[2][:, f0:f1], t.fm()[2][:, f0:f1], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(c.get_e(), t.e(n), rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(c.get_q(), t.q(n), rtol=1e-8, atol=1e-9)


@pytest.mark.parametrize("qfree", [True, False])
def test_sorted_onehot_binned_and_coop_levels(oracle, capi, monkeypatch, qfree):
    # user-sorted one-hot table: level 1 goes through the binned single-pass kernels (wave / workgroup /
    # co-resident long columns), in q-free and in q-cache form; a few users are long enough for k_long_coop
    if qfree:
        monkeypatch.setenv("MFM_QFREE", "1")
    monkeypatch.setenv("MFM_NO_SCATTER", "1")
    X, y, shapes = ds.onehot_mf(120000, 40, 300, seed=11, sort_by_user=True)
    gi = ds.group_index_from_shapes(shapes)
    t, c, _ = _pair(oracle, capi, X, y, gi, 4)
    assert c.plan_flags()["qfree"] == qfree
    drv = CapiGibbs(c, t.clone(), X.shape[0], gi)
    for it in range(3):
        t.step()
        drv.step()
    np.testing.assert_allclose(c.get_state()[2], t.fm()[2], rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(c.get_state()[1], t.fm()[1], rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(c.get_e(), t.e(X.shape[0]), rtol=1e-7, atol=1e-7)
    names = _timing_classes(c, drv)
    assert "sweep_V_coop" in names or "sweep_V_heavy" in names


def _timing_classes(c, drv):
    c.timing_enable(True)
    c.timing_reset()
    drv.c.sweep_V(0, 1, drv.alpha, drv.lam_V, drv.mu_V, np.zeros(c.D))
    names = set(c.timing())
    c.timing_enable(False)
    return names


@pytest.mark.parametrize("stream", [False, True, "split"])
@pytest.mark.parametrize("design", ["onehot", "multihot"])
def test_blocks_match_oracle_and_flat(oracle, capi, monkeypatch, design, stream):
    # tests/regression/test_block.py:80-149 on the device path + against the oracle
    # stream: the statistics / un-sync pass (FMTrainer.hpp:268-275, :401-417) streaming over the training rows with the
    # sums in an LDS table (k_unsync_stream, the form of blocks with few rows under a long table), else by block row
    # through the inverse map
    # "split": streaming un-sync of the rows (which also applies the previous block's re-sync) + read-only statistics through
    # the inverse map, the form of blocks with too many rows for the LDS table
    monkeypatch.setenv("MFM_UNSYNC_STREAM_FORCE", "1" if stream is True else "0")
    monkeypatch.setenv("MFM_UNSYNC_SPLIT_FORCE", "1" if stream == "split" else "0")
    if design == "onehot":
        main, X_flat, blocks, y, shapes = ds.block_design()
        rank = 2
    else:
        main, X_flat, blocks, y, shapes = ds.multihot_block_design()
        rank = 3
    gi = ds.group_index_from_shapes(shapes)
    kw = dict(fit_w0=False)
    tb, cb, _ = _pair(oracle, capi, main, y, gi, rank, blocks, **kw)
    tf, cf, _ = _pair(oracle, capi, X_flat, y, gi, rank, (), **kw)
    tb0 = tb.clone()
    db = CapiGibbs(cb, tb.clone(), main.shape[0], gi, fit_w0=False)
    df = CapiGibbs(cf, tf.clone(), main.shape[0], gi, fit_w0=False)
    for it in range(6):
        tb.step()
        db.step()
        df.step()
        w0, w, V = tb.fm()
        _, gw, gV = cb.get_state()
        _, fw, fV = cf.get_state()
        np.testing.assert_allclose(gV, V, rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(gw, w, rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(gV, fV, rtol=1e-7, atol=1e-8)  # blocked == flat on the device
        np.testing.assert_allclose(gw, fw, rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(cb.get_e(), tb.e(main.shape[0]), rtol=1e-7, atol=1e-7)
    if stream:  # the LDS sums are taken in a fixed order: a second chain from the same state is bit-identical
        _, c2, _ = _pair(oracle, capi, main, y, gi, rank, blocks, **kw)
        d2 = CapiGibbs(c2, tb0.clone(), main.shape[0], gi, fit_w0=False)
        for it in range(6):
            d2.step()
        assert np.array_equal(c2.get_state()[2], cb.get_state()[2]) and np.array_equal(c2.get_e(), cb.get_e())


@pytest.mark.parametrize("grid", [False, True])
@pytest.mark.parametrize("design", ["onehot", "multihot", "dense_main"])
def test_conflict_batched_chain(oracle, capi, monkeypatch, design, grid):
    # chains over state too large for LDS run conflict-batched (k_chain_batched: cold entries in parallel, hot rows
    # staged in LDS and walked in order); forced here on small designs: relation-block sweeps (64-byte records, w and
    # V) and a main table with dense columns (16-byte records). grid: the cold parts as grid launches (k_cb_*), the form
    # big relation blocks take
    monkeypatch.setenv("MFM_CHAIN_FORCE_BATCHED", "1")
    if grid:
        monkeypatch.setenv("MFM_CHAIN_GRID_MIN", "0")
    else:
        monkeypatch.setenv("MFM_NO_CHAIN_GRID", "1")
    if design == "dense_main":
        X, y = ds.middle_data()
        gi, blocks, rank, kw = np.zeros(X.shape[1], dtype=np.int32), (), 3, {}
    else:
        main, X_flat, blocks, y, shapes = ds.block_design() if design == "onehot" else ds.multihot_block_design()
        X, gi, rank, kw = main, ds.group_index_from_shapes(shapes), 3, dict(fit_w0=False)
    t, c, _ = _pair(oracle, capi, X, y, gi, rank, blocks, **kw)
    drv = CapiGibbs(c, t.clone(), X.shape[0], gi, **kw)
    for it in range(5):
        t.step()
        drv.step()
        np.testing.assert_allclose(c.get_state()[2], t.fm()[2], rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(c.get_state()[1], t.fm()[1], rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(c.get_e(), t.e(X.shape[0]), rtol=1e-7, atol=1e-7)


@pytest.mark.parametrize("per_row", [3, 6])
def test_conflict_batched_chain_many_batches(oracle, capi, monkeypatch, per_row):
    # the one-launch form of the grid-batched chains (k_cb_persist) classes a batch's cold entries by whether the batch before /
    # after touches their row and moves the "far" statistics / updates off the hand-over path; that only shows with three or
    # more batches. A hot-row cap of 64 cuts this 48-column main table into ~20-40 batches of one to three columns.
    monkeypatch.setenv("MFM_CHAIN_FORCE_BATCHED", "1")
    monkeypatch.setenv("MFM_CHAIN_GRID_MIN", "0")
    monkeypatch.setenv("MFM_CHAIN_HOT_CAP", "64")
    rng = np.random.default_rng(11 + per_row)
    n, d, rank = 6000, 48, 3
    cols = np.stack([rng.choice(d, size=per_row, replace=False) for _ in range(n)])
    vals = np.round(rng.uniform(-1.5, 1.5, size=(n, per_row)), 2)
    vals[vals == 0.0] = 0.5
    X = sps.csr_matrix((vals.ravel(), cols.ravel(), np.arange(0, n * per_row + 1, per_row)), shape=(n, d))
    X.sort_indices()
    y = ds.fm_score(X, 0.3, rng.normal(size=d) * 0.3, rng.normal(size=(d, rank)) * 0.3) + rng.normal(size=n) * 0.3
    gi = np.zeros(d, dtype=np.int32)
    t, c, _ = _pair(oracle, capi, X, y, gi, rank, ())
    drv = CapiGibbs(c, t.clone(), n, gi)
    for it in range(3):
        t.step()
        drv.step()
        np.testing.assert_allclose(c.get_state()[2], t.fm()[2], rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(c.get_state()[1], t.fm()[1], rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(c.get_e(), t.e(n), rtol=1e-7, atol=1e-7)


def test_block_only_design_no_main_columns(oracle, capi):
    # X=None => (N, 0) main table (base.py:230-233)
    main, X_flat, blocks, y, shapes = ds.multihot_block_design()
    empty = sps.csr_matrix((main.shape[0], 0))
    gi = ds.group_index_from_shapes(shapes[1:])
    t, c, _ = _pair(oracle, capi, empty, y, gi, 3, blocks)
    d = CapiGibbs(c, t.clone(), main.shape[0], gi)
    for _ in range(3):
        t.step()
        d.step()
    np.testing.assert_allclose(c.get_state()[2], t.fm()[2], rtol=1e-7, atol=1e-8)


def test_rank_zero(oracle, capi):
    X, score = ds.middle_data(300)
    y = score
    t, c, gi = _pair(oracle, capi, X, y, None, 0)
    d = CapiGibbs(c, t.clone(), X.shape[0], gi)
    for _ in range(3):
        t.step()
        d.step()
    np.testing.assert_allclose(c.get_state()[1], t.fm()[1], rtol=1e-8, atol=1e-9)


def test_predict_modes(oracle, capi):
    main, X_flat, blocks, y, shapes = ds.multihot_block_design()
    rng = np.random.default_rng(3)
    D = X_flat.shape[1]
    samples = [(rng.normal(), rng.normal(size=D) * 0.3, rng.normal(size=(D, 5)) * 0.3) for _ in range(4)]
    od = oracle.OracleDesign(main, blocks)
    scores = np.stack([od.predict_score(*s) for s in samples])
    dev = capi.Design(main, blocks)
    np.testing.assert_allclose(dev.predict(samples, 0), scores.mean(axis=0), rtol=1e-11, atol=1e-11)
    from scipy import special

    phi = (1 + special.erf(scores * np.sqrt(0.5))) / 2
    np.testing.assert_allclose(dev.predict(samples, 1), phi.mean(axis=0), rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(capi.Design(X_flat).predict(samples, 0), scores.mean(axis=0), rtol=1e-10, atol=1e-10)
    cuts = [np.sort(rng.normal(size=3)) for _ in samples]
    p = dev.predict(samples, 2, cuts)
    exp = np.zeros((main.shape[0], 4))
    for sc, cp in zip(scores, cuts):
        cdf = (1 + special.erf((cp[None, :] - sc[:, None]) * np.sqrt(0.5))) / 2
        full = np.hstack([np.zeros((sc.shape[0], 1)), cdf, np.ones((sc.shape[0], 1))])
        exp += full[:, 1:] - full[:, :-1]
    np.testing.assert_allclose(p, exp / len(samples), rtol=1e-10, atol=1e-12)


def test_sample_store_predicts_like_host_samples(oracle, capi):
    # device-resident posterior samples (mfm_store_*): pushed from the host and snapshotted from a training context;
    # Predictor::predict* over the store == over host copies of the same samples (all three modes), == the oracle
    main, X_flat, blocks, y, shapes = ds.multihot_block_design()
    rng = np.random.default_rng(4)
    D, K = X_flat.shape[1], 4
    samples = [(rng.normal(), rng.normal(size=D) * 0.3, rng.normal(size=(D, K)) * 0.3) for _ in range(5)]
    st = capi.Store(D, K)
    for s in samples[:3]:
        st.push(*s)
    c = capi.Context(main, y, blocks, rank=K, group_index=ds.group_index_from_shapes(shapes))
    for s in samples[3:]:
        c.set_state(*s)
        st.push_ctx(c)  # device-to-device snapshot of the live state
    assert len(st) == 5
    for k, s in enumerate(samples):
        w0, w, V = st.get(k)
        assert w0 == s[0] and np.array_equal(w, s[1]) and np.array_equal(V, s[2])
    dev = capi.Design(main, blocks)
    cuts = [np.sort(rng.normal(size=2)) for _ in samples]
    for mode in (0, 1, 2):
        a = st.predict(dev, mode, cuts if mode == 2 else None)
        b = dev.predict(samples, mode, cuts if mode == 2 else None)
        assert np.array_equal(a, b)
    od = oracle.OracleDesign(main, blocks)
    want = np.mean([od.predict_score(*s) for s in samples[1:4]], axis=0)
    np.testing.assert_allclose(st.predict(dev, 0, first=1, count=3), want, rtol=1e-11, atol=1e-11)
    with pytest.raises(ValueError):
        st.predict(dev, 0, first=3, count=5)


@pytest.mark.parametrize("K,unit", [(4, True), (7, False), (20, True), (0, True)])
def test_store_prediction_in_one_pass_over_the_rows(oracle, capi, monkeypatch, K, unit):
    # designs without relation blocks: Predictor::predict* over a store scores ALL samples in one pass over the test rows
    # (k_score_store: samples as the inner loop, sums in registers); bit-identical to the per-sample passes
    # (MFM_PREDICT_PER_SAMPLE=1) in all three modes, including sample sub-ranges, == the oracle's scorer
    X, y, shapes = ds.onehot_mf(5003, 120, 60, seed=3)
    if not unit:
        X = X.copy()
        X.data = np.where(np.arange(X.nnz) % 3 == 0, 0.5, 1.5)
    rng = np.random.default_rng(11)
    D, S = X.shape[1], 9
    samples = [(rng.normal(), rng.normal(size=D) * 0.3, rng.normal(size=(D, K)) * 0.3) for _ in range(S)]
    st = capi.Store(D, K)
    for smp in samples:
        st.push(*smp)
    dev = capi.Design(X, [])
    cuts = [np.sort(rng.normal(size=3)) for _ in samples]
    got = {}
    for mode in (0, 1, 2):
        got[mode] = st.predict(dev, mode, cuts if mode == 2 else None)
        got[mode, "sub"] = st.predict(dev, mode, cuts[2:7] if mode == 2 else None, first=2, count=5)
    monkeypatch.setenv("MFM_PREDICT_PER_SAMPLE", "1")
    for mode in (0, 1, 2):
        assert np.array_equal(got[mode], st.predict(dev, mode, cuts if mode == 2 else None))
        assert np.array_equal(got[mode, "sub"], st.predict(dev, mode, cuts[2:7] if mode == 2 else None, first=2, count=5))
        assert np.array_equal(got[mode], dev.predict(samples, mode, cuts if mode == 2 else None))
    od = oracle.OracleDesign(X, [])
    want = np.mean([od.predict_score(*smp) for smp in samples], axis=0)
    np.testing.assert_allclose(got[0], want, rtol=1e-11, atol=1e-11)
    assert got[2].shape == (5003, 4) and np.allclose(got[2].sum(axis=1), 1.0)


def test_more_than_sixteen_relation_blocks(oracle, capi):
    # BaseFMTrainer.hpp:58-68 takes any vector of relation blocks; the first 16 travel in the kernel arguments of the q-cache
    # build and of the scorer, the rest through device arrays: 19 small blocks against the oracle (two full iterations) and
    # against the oracle's scorer through a prediction design
    rng = np.random.default_rng(12)
    n, n_blocks, K = 2500, 19, 3
    main = sps.csr_matrix(rng.normal(size=(n, 2)))
    blocks, shapes = [], [2]
    for b in range(n_blocks):
        rows_b = 15 + b
        B = sps.hstack([sps.identity(rows_b), sps.csr_matrix(rng.normal(size=(rows_b, 1)))]).tocsr()
        blocks.append((rng.integers(0, rows_b, size=n), B))
        shapes.append(B.shape[1])
    y = rng.normal(size=n)
    gi = ds.group_index_from_shapes(shapes)
    t, c, _ = _pair(oracle, capi, main, y, gi, K, blocks=blocks)
    drv = CapiGibbs(c, t.clone(), n, gi)
    for it in range(2):
        t.step()
        drv.step()
    w0, w, V = t.fm()
    gw0, gw, gV = c.get_state()
    np.testing.assert_allclose(gV, V, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(gw, w, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(c.get_e(), t.e(n), rtol=1e-7, atol=1e-8)
    dev = capi.Design(main, blocks)
    got = dev.predict([(w0, w, V)], 0, None)
    np.testing.assert_allclose(got, oracle.OracleDesign(main, blocks).predict_score(w0, w, V), rtol=1e-10, atol=1e-10)


def test_error_paths(capi):
    X, y = ds.toy()
    with pytest.raises(RuntimeError, match="index mapping points to non-existing row"):
        capi.Context(X, y, [(np.array([0, 1, 2, 7]), sps.csr_matrix(np.eye(3)))], rank=2, group_index=np.zeros(12, np.int32))
    with pytest.raises(ValueError, match="No matching index for group index"):
        capi.Context(X, y, rank=2, group_index=np.array([0, 0, 0, 2, 2, 2, 2, 2, 2], np.int32))
    with pytest.raises(ValueError):
        capi.Context(X, y, rank=2, group_index=np.zeros(5, np.int32))


def test_slot_order_scorer_residual_and_sums(capi, monkeypatch):
    # update_e of a regression on a table that takes the persistent sweep: the residual is scored straight in the sweep's slot
    # order (k_res_score) with sum e / sum e^2 as by-products, and moved to row order only on request. Held against numpy on
    # the state the context reports, before and after a further fused sweep reads it back in slot order.
    monkeypatch.setenv("MFM_SCATTER_MIN_NNZ", "1000")
    n, K = 90001, 6
    X, y, shapes = ds.onehot_mf(n, 230, 170, seed=12, sort_by_user=True)
    gi = ds.group_index_from_shapes(shapes)
    c = capi.Context(X, y, rank=K, group_index=gi)
    assert c.plan_flags()["resident"]
    D, G = X.shape[1], int(max(gi)) + 1
    rng = np.random.default_rng(3)
    c.set_state(0.3, rng.normal(size=D) * 0.1, rng.normal(size=(D, K)) * 0.1)
    u, i = X.indices[0::2], X.indices[1::2]

    def numpy_residual():
        w0, w, V = c.get_state()
        return w0 + w[u] + w[i] + np.einsum("nf,nf->n", V[u], V[i]) - y

    c.update_e_regression()
    s, s2 = c.reduce_e()  # (materialises the row order)
    want = numpy_residual()
    np.testing.assert_allclose(c.get_e(), want, rtol=1e-12, atol=1e-12)
    assert abs(s - want.sum()) < 1e-8 * n and abs(s2 - (want ** 2).sum()) < 1e-8 * n
    # a fused sweep that starts from the slot-ordered residual, against the same sweep started from the row-ordered one
    lam_w, mu_w = np.full(G, 1.5), np.zeros(G)
    lam_V, mu_V = np.full((G, K), 2.0), np.zeros((G, K))
    zw, zv = rng.normal(size=D), rng.normal(size=(K, D))
    state = c.get_state()
    c.update_e_regression()  # residual in slot order now
    c.sweep_wV(1.1, 0.05, lam_w, mu_w, zw, 0, K, lam_V, mu_V, zv)
    a_state, a_e = c.get_state(), c.get_e()
    c.set_state(*state)
    c.update_e_regression()
    c.get_e()  # row order
    c.sweep_wV(1.1, 0.05, lam_w, mu_w, zw, 0, K, lam_V, mu_V, zv)
    b_state, b_e = c.get_state(), c.get_e()
    assert a_state[0] == b_state[0]
    np.testing.assert_array_equal(a_state[1], b_state[1])
    np.testing.assert_array_equal(a_state[2], b_state[2])
    np.testing.assert_array_equal(a_e, b_e)


@pytest.mark.parametrize("cus,store", [(2, True), (2, False), (3, True)])
def test_resident_sweep_beyond_the_on_chip_capacity(oracle, capi, monkeypatch, cus, store):
    # Tables with more rows per CU than the 512 x 80 on-chip slots (the 10.48 M-row cliff of round 4): the persistent sweep keeps
    # 80 slots per thread on chip and streams the others' residual from / to the slot-ordered buffer every sweep
    # (k_mf_resident<.., OVF>, ResPlan::RX). Here 120 001 rows on 2 (3) "CUs": 60 000 (40 001 + ...) rows per workgroup. The chain of
    # mfm_sweep_wV + update_e against the oracle draw for draw, with the residual written back (ctypes default) and with the
    # trainer's policy (not written back: recomputed on demand), and the residual itself after the last sweep.
    monkeypatch.setenv("MFM_SCATTER_MIN_NNZ", "1000")
    monkeypatch.setenv("MFM_RES_CUS", str(cus))
    n = 120001 if cus == 2 else 150001
    X, y, shapes = ds.onehot_mf(n, 260, 140, seed=5, sort_by_user=True)
    gi = ds.group_index_from_shapes(shapes)
    t, c, _ = _pair(oracle, capi, X, y, gi, 4)
    assert c.plan_flags()["resident"] and c.plan_flags()["resident_overflow"]
    if not store:
        c.set_residual_policy(True)
    drv = CapiGibbs(c, t.clone(), n, gi, fused=True)
    seen = {}
    for it in range(3):
        t.step()
        drv.step(before_update_e=lambda: seen.update(e=c.get_e()))
        np.testing.assert_allclose(c.get_state()[2], t.fm()[2], rtol=1e-7, atol=1e-8, err_msg="iteration %d" % it)
        np.testing.assert_allclose(c.get_state()[1], t.fm()[1], rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(seen["e"], c.get_e(), rtol=0, atol=1e-9)  # the sweeps' residual == the recomputed one
    np.testing.assert_allclose(c.get_e(), t.e(n), rtol=1e-7, atol=1e-7)
    assert "sweep_V_resident" in _timing_classes(c, drv)
    # separate sweeps (mfm_sweep_w, then mfm_sweep_V factor by factor) take the same path
    t2, c2, _ = _pair(oracle, capi, X, y, gi, 4)
    drv2 = CapiGibbs(c2, t2.clone(), n, gi, fused=False)
    t2.step()
    drv2.step()
    np.testing.assert_allclose(c2.get_state()[2], t2.fm()[2], rtol=1e-7, atol=1e-8)
