"""Kernel-level parity of the classification / ordered-probit pieces of update_e (FMTrainer.hpp:498-521), through the
C ABI:

  * the device erfcx against OUTPUTS OF THE REFERENCE'S OWN Faddeeva.cc (tests/golden/faddeeva_erfcx*.npz);
  * mfm_oprobit_eval (log-likelihood, d/dgamma, gamma-space Hessian: OProbitSampler.hpp:402-413, :111-236) against the
    oracle's row loop on the same scores, on every branch of safe_lcdf / safe_lccdf / safe_ldiff, for the whole table
    and for a row subset (cutpoint groups, BaseFMTrainer.hpp:79-104);
  * the device truncated-normal samplers (util.hpp:15-78) per bound class against the oracle's (libstdc++ mt19937)
    samplers and the analytic distribution: the device draws come from per-row Philox streams, so parity is
    distributional -- two-sample and one-sample Kolmogorov-Smirnov distances and the first two moments on 1e6 draws;
  * mfm_design_score_ctx (the scorer of the LibFM-style callbacks, utils/callbacks/libfm.py:85) against the oracle.
"""
import os

import numpy as np
import pytest
import scipy.sparse as sps
from scipy import stats

from . import datasets as ds

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def capi():
    from myfm_amd import _capi

    if _capi.lib().mfm_device_count() < 1:
        pytest.fail("no HIP device visible: the gpu tests need a real MI355X")
    return _capi


@pytest.mark.parametrize("name", ["faddeeva_erfcx", "faddeeva_erfcx_wide"])
def test_device_erfcx_matches_reference_faddeeva_vectors(capi, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    got = capi.device_erfcx(g["x"])
    fin = np.isfinite(g["erfcx"])
    assert fin.sum() >= 200
    # the device evaluates exp(x^2) erfc(x) below 3 and a continued fraction above, with the GPU's libm: a few ulp
    np.testing.assert_allclose(got[fin], g["erfcx"][fin], rtol=2e-13, atol=0)
    assert np.all(np.isinf(got[~fin]))


def _ordered_problem(n, n_class, seed):
    rng = np.random.default_rng(seed)
    X = sps.csr_matrix(rng.normal(size=(n, 1)))
    y = rng.integers(0, n_class, size=n).astype(np.float64)
    # cutpoints about 0.7 apart; scores reach 5 beyond the outermost ones, so that every label meets arguments on both
    # sides of zero: all three branches of safe_ldiff (y > 0 | x < 0 | straddling 0) and both branches of safe_lcdf
    # (x > 1) / safe_lccdf (x > -1) occur
    scores = rng.uniform(-5.0, 0.7 * (n_class - 2) + 5.0, size=n)
    return X, y, scores


# (32 classes: the last count with the accumulators in LDS; 33 and 70: in global memory -- OProbitSampler.hpp:36-46 has no bound)
@pytest.mark.parametrize("n_class", [2, 3, 5, 9, 17, 32, 33, 70])
def test_oprobit_eval_matches_oracle(capi, oracle, n_class):
    n = 20000
    X, y, scores = _ordered_problem(n, n_class, seed=n_class)
    c = capi.Context(X, y, rank=0)
    c.set_e(scores)
    g_all = c.oprobit_add_group(n_class)
    rng = np.random.default_rng(1)
    for trial in range(3):
        alpha = np.concatenate([[rng.normal() * 0.3], np.log(0.7) + rng.normal(size=n_class - 2) * 0.2 * trial])
        gamma = np.concatenate([[alpha[0]], alpha[0] + np.cumsum(np.exp(alpha[1:]))])  # OProbitSampler.hpp:95-101
        want = oracle.oprobit_eval(n_class, alpha, scores, y)
        ll, dg, H = c.oprobit_eval(g_all, gamma)
        # sums of 20 000 terms in a different (fixed) order, device libm vs glibc: 1e-10 relative to the term scale
        assert abs(ll - want["ll"]) <= 1e-10 * abs(want["ll"])
        np.testing.assert_allclose(dg, want["dgamma"], rtol=1e-10, atol=1e-10 * np.abs(want["dgamma"]).max())
        np.testing.assert_allclose(H, want["Hg"], rtol=1e-10, atol=1e-10 * np.abs(want["Hg"]).max())
        ll2, dg2, _ = c.oprobit_eval(g_all, gamma, want_h=False)
        assert ll2 == ll and np.array_equal(dg2, dg)
        # every branch was exercised
        x_hi = gamma[np.minimum(y.astype(int), n_class - 2)] - scores
        mid = (y > 0) & (y < n_class - 1)
        if mid.any():
            x_lo = gamma[np.maximum(y.astype(int) - 1, 0)] - scores
            assert (x_lo[mid] > 0).any() and (x_hi[mid] < 0).any() and ((x_lo[mid] <= 0) & (x_hi[mid] >= 0)).any()
        assert (x_hi[y == 0] > 1).any() and (x_hi[y == 0] <= 1).any()
        top = y == n_class - 1
        assert (x_hi[top] > -1).any() and (x_hi[top] <= -1).any()


def test_oprobit_eval_row_subsets(capi, oracle):
    # two cutpoint groups with different class counts over disjoint row sets (FMLearningConfig.hpp:15)
    n = 12000
    rng = np.random.default_rng(7)
    X = sps.csr_matrix(rng.normal(size=(n, 1)))
    rows_a = np.sort(rng.choice(n, size=n // 3, replace=False))
    rows_b = np.setdiff1d(np.arange(n), rows_a)
    y = np.zeros(n)
    y[rows_a] = rng.integers(0, 3, size=rows_a.size)
    y[rows_b] = rng.integers(0, 6, size=rows_b.size)
    scores = rng.normal(size=n) * 2.5
    c = capi.Context(X, y, rank=0)
    c.set_e(scores)
    ga, gb = c.oprobit_add_group(3, rows_a), c.oprobit_add_group(6, rows_b)
    for g, rows, C_ in ((ga, rows_a, 3), (gb, rows_b, 6)):
        alpha = rng.normal(size=C_ - 1) * 0.4
        gamma = np.concatenate([[alpha[0]], alpha[0] + np.cumsum(np.exp(alpha[1:]))])
        want = oracle.oprobit_eval(C_, alpha, scores, y, rows=rows)
        ll, dg, H = c.oprobit_eval(g, gamma)
        assert abs(ll - want["ll"]) <= 1e-10 * abs(want["ll"])
        np.testing.assert_allclose(dg, want["dgamma"], rtol=1e-10, atol=1e-10 * np.abs(want["dgamma"]).max())
        np.testing.assert_allclose(H, want["Hg"], rtol=1e-10, atol=1e-10 * np.abs(want["Hg"]).max())


# util.hpp:15-60: one case per branch of the samplers
TN_CASES = [
    ("left", -1.3, None),     # bound < 0: rejection from N(0, 1)
    ("left", 0.0, None),      # bound >= 0: exponential proposal (Robert 2009)
    ("left", 2.5, None),
    ("right", None, 0.7),     # = -left(-hi)
    ("right", None, -1.8),
    ("twoside", -0.8, 1.1),   # contains 0
    ("twoside", -3.0, -1.2),  # both negative
    ("twoside", 0.9, 2.4),    # both positive
]


@pytest.mark.parametrize("kind,lo,hi", TN_CASES)
def test_truncated_normal_sampler_distribution(capi, oracle, kind, lo, hi):
    n = 1_000_000
    a = -np.inf if lo is None else lo
    b = np.inf if hi is None else hi
    dev = capi.device_truncated_normal(kind, 0.0 if lo is None else lo, 0.0 if hi is None else hi, n, seed=11, draw_index=3)
    if kind == "left":
        ref = oracle.tn_left_many(5, lo, n)
    elif kind == "right":
        ref = -oracle.tn_left_many(5, -hi, n)  # util.hpp:68-71
    else:
        ref = oracle.tn_twoside_many(5, lo, hi, n)
    assert np.all(dev > a) and np.all(dev < b) and np.all(np.isfinite(dev))
    dist = stats.truncnorm(a, b)
    # one-sample KS distance to the analytic law: P(sqrt(n) D > 2.2) ~ 1e-4
    for x in (dev, ref):
        d = stats.kstest(x, dist.cdf).statistic
        assert d < 2.2 / np.sqrt(n), d
    # two-sample distance device vs oracle sampler
    assert stats.ks_2samp(dev, ref).statistic < 2.2 * np.sqrt(2.0 / n)
    # first two moments within 5 standard errors of the analytic values
    m, v = dist.mean(), dist.var()
    assert abs(dev.mean() - m) < 5 * np.sqrt(v / n)
    kurt_term = dist.moment(4) - 4 * m * dist.moment(3) + 6 * m * m * dist.moment(2) - 3 * m ** 4  # E (x - m)^4
    assert abs(dev.var() - v) < 5 * np.sqrt((kurt_term - v * v) / n)
    # reproducible for a key, different for another draw index
    again = capi.device_truncated_normal(kind, 0.0 if lo is None else lo, 0.0 if hi is None else hi, 1000, seed=11, draw_index=3)
    other = capi.device_truncated_normal(kind, 0.0 if lo is None else lo, 0.0 if hi is None else hi, 1000, seed=11, draw_index=4)
    assert np.array_equal(again, dev[:1000]) and not np.array_equal(other, again)


@pytest.mark.parametrize("design", ["onehot", "blocks"])
def test_score_ctx_matches_oracle(capi, oracle, design):
    # FM::predict_score of the live sample (FMTrainer.hpp:78) on a test design: device-resident state vs oracle scorer
    rng = np.random.default_rng(5)
    if design == "onehot":
        X, y, shapes = ds.onehot_mf(6000, 80, 50, seed=5)
        blocks, K = (), 6
        Xt, _, _ = ds.onehot_mf(1500, 80, 50, seed=6, sort_by_user=False)
        tblocks = ()
    else:
        X, _, blocks, y, shapes = ds.multihot_block_design()
        K = 3
        ui, ii = rng.integers(0, 12, size=300), rng.integers(0, 9, size=300)
        Xt = sps.random(300, 4, density=0.6, random_state=np.random.RandomState(9), format="csr")
        tblocks = [(ui.astype(np.int64), blocks[0][1]), (ii.astype(np.int64), blocks[1][1])]
    gi = ds.group_index_from_shapes(shapes)
    c = capi.Context(X, y, blocks, rank=K, group_index=gi)
    D = c.D
    w0, w, V = rng.normal(), rng.normal(size=D) * 0.3, rng.normal(size=(D, K)) * 0.3
    c.set_state(w0, w, V)
    c.set_w0(w0)
    dev = capi.Design(Xt, tblocks)
    want = oracle.OracleDesign(Xt, tblocks).predict_score(w0, w, V)
    np.testing.assert_allclose(dev.score_ctx(c), want, rtol=1e-11, atol=1e-11)
    # ... and after a sweep moved the resident state
    t = oracle.OracleTrainer(X, y, blocks, rank=K, group_index=gi)
    t.set_fm(w0, w, V)
    c.update_e_regression()
    z = rng.normal(size=D)
    lam, mu = np.full(int(gi.max()) + 1, 0.5), np.zeros(int(gi.max()) + 1)
    c.sweep_w(1.3, lam, mu, z)
    _, w2, V2 = c.get_state()
    want2 = oracle.OracleDesign(Xt, tblocks).predict_score(w0, w2, V2)
    np.testing.assert_allclose(dev.score_ctx(c), want2, rtol=1e-11, atol=1e-11)
