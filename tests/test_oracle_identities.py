"""Pins the CPU oracle with the identities the reference's own tests assert.

The reference cannot be built here (Eigen absent), so these are the checks available:
closed-form score, flat == blocked, statistical recovery, erfcx vs the real Faddeeva.cc.
"""
import numpy as np
import pytest
import scipy.sparse as sps

from . import datasets as ds


def test_predict_score_closed_form(oracle):
    # tests/test_utils.py:16-25
    X, _ = ds.middle_data(200)
    rng = np.random.default_rng(0)
    w0, w, V = 0.7, rng.normal(size=3), rng.normal(size=(3, 4))
    got = oracle.OracleDesign(X).predict_score(w0, w, V)
    np.testing.assert_allclose(got, ds.fm_score(X, w0, w, V), rtol=1e-12, atol=1e-12)


def test_predict_score_blocks_equals_flat(oracle):
    main, X_flat, blocks, y, _ = ds.multihot_block_design()
    rng = np.random.default_rng(1)
    D = X_flat.shape[1]
    w0, w, V = -0.2, rng.normal(size=D), rng.normal(size=(D, 3))
    a = oracle.OracleDesign(X_flat).predict_score(w0, w, V)
    b = oracle.OracleDesign(main, blocks).predict_score(w0, w, V)
    np.testing.assert_allclose(a, b, rtol=1e-11, atol=1e-11)


@pytest.mark.parametrize("design", ["onehot", "multihot"])
def test_flat_equals_blocked(oracle, design):
    # tests/regression/test_block.py:80-149: every kept sample's V matches between the flat
    # hstack design and the RelationBlock design under one seed (assert_allclose default 1e-7).
    if design == "onehot":
        main, X_flat, blocks, y, shapes = ds.block_design()
        rank = 2
    else:
        main, X_flat, blocks, y, shapes = ds.multihot_block_design()
        rank = 3
    gi = ds.group_index_from_shapes(shapes)
    kw = dict(rank=rank, fit_w0=False, group_index=gi, n_iter=30, n_kept_samples=30)
    s_flat, h_flat, _ = oracle.fit(X_flat, y, **kw)
    s_blk, h_blk, _ = oracle.fit(main, y, blocks, **kw)
    assert len(s_flat) == len(s_blk) == 30
    for a, b in zip(s_flat, s_blk):
        np.testing.assert_allclose(a[2], b[2], rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(a[1], b[1], rtol=1e-7, atol=1e-9)
    # doc/source/relation-blocks.rst:205-210
    assert max(np.abs(a[1] - b[1]).max() for a, b in zip(s_flat[:3], s_blk[:3])) < 1e-5
    # predictions agree across formats (test_block.py:147-149)
    pf = np.mean([oracle.OracleDesign(main, blocks).predict_score(*s) for s in s_flat], axis=0)
    pb = np.mean([oracle.OracleDesign(X_flat).predict_score(*s) for s in s_blk], axis=0)
    np.testing.assert_allclose(pf, pb, rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize("alpha_inv", [0.3, 1.0, 3])
def test_regression_recovery(oracle, alpha_inv):
    # tests/regression/test_fit.py:19-72
    X, score = ds.middle_data()
    rns = np.random.RandomState(0)
    y = score + alpha_inv * rns.normal(0, 1, size=score.shape)
    samples, hypers, _ = oracle.fit(X, y, rank=3, n_iter=100, n_kept_samples=100)
    last_alpha = np.array([h["alpha"] for h in hypers[-20:]])
    assert np.all(last_alpha > (1 / alpha_inv ** 2) / 2)
    assert np.all(last_alpha < (1 / alpha_inv ** 2) * 2)
    for w0, w, V in samples[-20:]:
        assert abs(w0 - ds.STUB_W0) < 0.5
        assert np.all(np.abs(w - ds.STUB_W) < 1.0)
        for i in range(3):
            for j in range(i + 1, 3):
                cross = ds.STUB_V[:, i].dot(ds.STUB_V[:, j])
                if abs(cross) < 0.1:
                    continue
                sign = cross / abs(cross)
                c = V[i].dot(V[j])
                assert c > sign * cross * 0.5 and c < sign * cross * 2


def test_classification_recovery(oracle):
    # tests/classification/test_classification.py:13-70
    X, score = ds.middle_data()
    rns = np.random.RandomState(0)
    sn = score + rns.normal(0, 1, size=score.shape)
    sn -= sn.mean()
    y = (sn > 0).astype(np.float64) * 2 - 1  # base.py:385-386
    samples, _, _ = oracle.fit(X, y, rank=3, n_iter=200, n_kept_samples=200, task=oracle.CLASSIFICATION)
    for w0, w, V in samples[-20:]:
        for i in range(3):
            for j in range(i + 1, 3):
                cross = ds.STUB_V[:, i].dot(ds.STUB_V[:, j])
                if abs(cross) < 0.5:
                    continue
                sign = cross / abs(cross)
                c = V[i].dot(V[j])
                assert c > sign * cross * 0.5 and c < sign * cross * 2


def test_oprobit_recovery(oracle):
    # tests/oprobit/test_oprobit_1dim.py:10-61
    n = 1000
    cps = np.array([0.0, 0.5, 1.5])
    rns = np.random.RandomState(0)
    x = rns.normal(0, 2, size=n)
    y = np.zeros(n)
    score = x * 0.5 + rns.randn(n)
    for c in cps:
        y += (score > c).astype(np.int64)
    samples, _, t = oracle.fit(
        sps.csr_matrix(x[:, None]), y, rank=0, fit_w0=False, n_iter=100, n_kept_samples=100, task=oracle.ORDERED
    )
    for s in samples[-10:]:
        c1, c2, c3 = s[3]
        assert abs(c1) < 0.25 and abs(c2 - c1 - 0.5) < 0.25 and abs(c3 - c1 - 1.5) < 0.25
    assert t.mh_accept(0) > 20


def test_erfcx_against_reference_faddeeva(oracle):
    R = oracle.ref_faddeeva()
    if R is None:
        pytest.skip("oracle/_ref/libfaddeeva_ref.so not built (reference absent)")
    xs = np.concatenate([np.linspace(-6, 60, 4001), 10 ** np.linspace(-8, 9, 300), -(10 ** np.linspace(-8, 0.7, 100))])
    err = max(abs(oracle.lib().orc_erfcx(float(x)) / R.ref_erfcx(float(x)) - 1) for x in xs)
    assert err < 5e-15


def test_truncated_normal_moments(oracle):
    # util.hpp:15-60: samples respect the truncation and have the right mean
    import ctypes as C
    from scipy import stats

    L = oracle.lib()
    for mu_minus in [-1.5, 0.0, 0.8, 3.0]:
        out = np.empty(20000)
        L.orc_tn_left_many(7, mu_minus, out.size, out.ctypes.data_as(C.c_void_p))
        assert out.min() > mu_minus
        expect = stats.truncnorm(mu_minus, np.inf).mean()
        assert abs(out.mean() - expect) < 0.03
    for lo, hi in [(-1.0, 0.5), (0.5, 1.2), (-2.5, -1.0)]:
        out = np.empty(20000)
        L.orc_tn_twoside_many(11, lo, hi, out.size, out.ctypes.data_as(C.c_void_p))
        assert out.min() >= lo and out.max() <= hi
        assert abs(out.mean() - stats.truncnorm(lo, hi).mean()) < 0.02


def test_toy_config1(oracle):
    # BASELINE config 1: MyFMRegressor rank=4 (and examples/toy.py's classifier) on the 4x9 toy.
    X, y = ds.toy()
    s, h, _ = oracle.fit(X, y, rank=4, n_iter=100)
    assert len(s) == 95 and len(h) == 100 and np.isfinite(s[-1][2]).all()
    s, h, _ = oracle.fit(X, y * 2 - 1, rank=4, n_iter=100, task=oracle.CLASSIFICATION)
    assert np.isfinite(s[-1][2]).all()
