"""End-to-end GPU tests through the drop-in boundary (myfm_amd estimators -> _myfm -> C ABI).

They read like the reference's own tests (tests/regression/test_block.py, test_fit.py,
tests/classification/test_classification.py, tests/oprobit/test_oprobit_1dim.py) plus draw-for-draw
comparisons with the CPU oracle where the chain is seed-reproducible (regression).
"""
import pickle
import tempfile

import numpy as np
import pytest
import scipy.sparse as sps

from . import datasets as ds

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def myfm():
    import myfm_amd
    from myfm_amd import _myfm

    if _myfm.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu tests need a real MI355X")
    return myfm_amd


class RunningMean:
    """what RegressionCallback / ClassificationCallback accumulate (utils/callbacks/libfm.py:82-113)."""

    def __init__(self, X, transform=lambda s: s, X_rel=()):
        self.X, self.X_rel, self.f = X, list(X_rel), transform
        self.sum, self.n = None, 0

    def __call__(self, i, fm, hyper, history):
        p = self.f(fm.predict_score(self.X, self.X_rel))
        self.sum = p if self.sum is None else self.sum + p
        self.n += 1
        return False, None


def test_libfm_callbacks(myfm, oracle):
    # utils/callbacks/libfm.py semantics: running means of per-iteration test scores; the live sample is
    # scored on the device from the resident state (same numbers as scoring the downloaded sample)
    from myfm_amd.utils import ClassificationCallback, OrderedProbitCallback, RegressionCallback

    X, score = ds.middle_data()
    y = score + np.random.RandomState(0).normal(0, 1, size=score.shape)
    cb = RegressionCallback(40, X_test=X, y_test=y, clip_min=-50, clip_max=50)
    fm = myfm.MyFMGibbsRegressor(3).fit(X, y, n_iter=40, n_kept_samples=40, callback=cb)
    np.testing.assert_allclose(fm.predict(X), cb.predictions / 40)
    assert list(cb.result_trace[-1]) == ["alpha", "rmse", "rmse_this", "rmse_all_but_5"]
    assert np.isnan(cb.result_trace[2]["rmse_all_but_5"]) and cb.result_trace[-1]["rmse"] < 1.5
    manual = np.mean([s.predict_score(X, []) for s in fm.predictor_.samples[5:]], axis=0)
    np.testing.assert_allclose(cb.prediction_all_but_5 / 35, manual)
    # relation blocks through the callback
    main, X_flat, blocks, yb, shapes = ds.multihot_block_design()
    rbs = [myfm.RelationBlock(list(m), b) for m, b in blocks]
    cb = RegressionCallback(10, X_test=main, y_test=yb, X_rel_test=rbs)
    fm = myfm.MyFMRegressor(3).fit(main, yb, rbs, group_shapes=shapes, n_iter=10, n_kept_samples=10, callback=cb)
    np.testing.assert_allclose(fm.predict(main, rbs), cb.predictions / 10)
    yc = y > np.median(y)
    cb = ClassificationCallback(30, X, yc)
    clf = myfm.MyFMGibbsClassifier(3).fit(X, yc, n_iter=30, n_kept_samples=30, callback=cb)
    np.testing.assert_allclose(clf.predict_proba(X), cb.predictions / 30)
    assert list(cb.result_trace[-1])[:3] == ["log_loss", "log_loss_this", "log_loss_all_but_5"]
    yo = np.digitize(y, np.quantile(y, [0.25, 0.5, 0.75])).astype(float)
    cb = OrderedProbitCallback(20, X_test=X, y_test=yo, n_class=4)
    op = myfm.MyFMOrderedProbit(2).fit(X, yo, n_iter=20, n_kept_samples=20, callback=cb)
    np.testing.assert_allclose(cb.predictions / 20, op.predict_proba(X))
    assert cb.result_trace[-1]["accuracy"] > 0.3


def test_kept_samples_stay_on_device_until_asked_for(myfm, oracle, monkeypatch):
    # FMTrainer.hpp:71-74 retention without host copies: predict() reads the samples in place; w_samples / V_samples,
    # pickling and per-sample predict_score materialise them; a sample the user edits detaches from the store
    X, y, shapes = ds.onehot_mf(20000, 300, 60, seed=8)
    gi = ds.group_index_from_shapes(shapes)
    fm = myfm.MyFMRegressor(5).fit(X, y, group_shapes=shapes, n_iter=8, n_kept_samples=6)
    Xt = X[::7]
    p_dev = fm.predict(Xt)
    samples, _, _ = oracle.fit(X, y, rank=5, group_index=gi, n_iter=8, n_kept_samples=6)
    want = np.mean([oracle.OracleDesign(Xt).predict_score(*s) for s in samples], axis=0)
    np.testing.assert_allclose(p_dev, want, rtol=1e-7, atol=1e-7)
    np.testing.assert_allclose(fm.V_samples[-1], samples[-1][2], rtol=1e-7, atol=1e-7)
    np.testing.assert_allclose(fm.w0_samples, [s[0] for s in samples], rtol=1e-9)
    blob = pickle.dumps(fm.predictor_)
    p2 = pickle.loads(blob)
    np.testing.assert_allclose(p2.predict(Xt, []), p_dev, rtol=1e-12, atol=1e-12)
    # host-sample mode gives the same numbers
    monkeypatch.setenv("MYFM_AMD_HOST_SAMPLES", "1")
    fm_h = myfm.MyFMRegressor(5).fit(X, y, group_shapes=shapes, n_iter=8, n_kept_samples=6)
    assert np.array_equal(fm_h.predict(Xt), p_dev)
    assert np.array_equal(fm_h.V_samples, fm.V_samples)
    # ... and so does a store whose reservation is refused (more than MFM_STORE_MAX_FRACTION of the free HBM): host copies
    monkeypatch.delenv("MYFM_AMD_HOST_SAMPLES")
    monkeypatch.setenv("MFM_STORE_MAX_FRACTION", "1e-12")
    fm_r = myfm.MyFMRegressor(5).fit(X, y, group_shapes=shapes, n_iter=8, n_kept_samples=6)
    assert np.array_equal(fm_r.predict(Xt), p_dev)
    assert np.array_equal(fm_r.V_samples, fm.V_samples)


def test_toy_config1(myfm):
    # BASELINE config 1 / examples/toy.py
    X, y = ds.toy()
    fm = myfm.MyFMRegressor(rank=4).fit(X, y, n_iter=100)
    assert fm.predict(X).shape == (4,) and len(fm.predictor_.samples) == 95
    clf = myfm.MyFMClassifier(rank=4).fit(X, y)
    p = clf.predict_proba(sps.csr_matrix(np.array([[24.0, 1, 0, 0, 0, 1, 0, 0, 0]])))
    assert p.shape == (1,) and 0 <= p[0] <= 1


def test_regression_matches_oracle_sample_by_sample(myfm, oracle):
    X, y, shapes = ds.onehot_mf(30000, 800, 150, seed=4)
    fm = myfm.MyFMRegressor(6, random_seed=7).fit(X, y, group_shapes=shapes, n_iter=10, n_kept_samples=10)
    samples, hypers, _ = oracle.fit(X, y, rank=6, seed=7, group_index=ds.group_index_from_shapes(shapes), n_iter=10,
                                    n_kept_samples=10)
    for it, (got, want) in enumerate(zip(fm.predictor_.samples, samples)):
        tol = 1e-9 if it == 0 else 1e-6
        assert abs(got.w0 - want[0]) < tol
        np.testing.assert_allclose(got.w, want[1], rtol=tol, atol=tol)
        np.testing.assert_allclose(got.V, want[2], rtol=tol, atol=tol)
    trace = fm.get_hyper_trace()
    np.testing.assert_allclose(trace["alpha"].values, [h["alpha"] for h in hypers], rtol=1e-6)
    np.testing.assert_allclose(trace["lambda_V[1,3]"].values, [h["lambda_V"][1, 3] for h in hypers], rtol=1e-6)
    # posterior-mean parity (north_star: RMSE <= 1e-3)
    pred = fm.predict(X[:5000])
    want = np.mean([oracle.OracleDesign(X[:5000]).predict_score(*s) for s in samples], axis=0)
    assert np.sqrt(np.mean((pred - want) ** 2)) < 1e-6


def test_rows_in_any_order(myfm, monkeypatch):
    # fit() sorts the rows by their first stored column for the device (the sampler is invariant to the row order);
    # a shuffled table, with a relation block riding along, gives the chain of the sorted one
    X, y, shapes = ds.onehot_mf(20000, 300, 60, seed=12, sort_by_user=True)
    rng = np.random.default_rng(0)
    side = sps.csr_matrix(rng.normal(size=(60, 2)))
    item = X.indices[1::2] - 300
    p = rng.permutation(X.shape[0])
    gs = shapes + [2]
    kw = dict(group_shapes=gs, n_iter=6, n_kept_samples=6)
    a = myfm.MyFMRegressor(4, random_seed=3).fit(X, y, [myfm.RelationBlock([int(v) for v in item], side)], **kw)
    b = myfm.MyFMRegressor(4, random_seed=3).fit(X[p], y[p], [myfm.RelationBlock([int(v) for v in item[p]], side)], **kw)
    monkeypatch.setenv("MYFM_AMD_KEEP_ROW_ORDER", "1")
    c = myfm.MyFMRegressor(4, random_seed=3).fit(X[p], y[p], [myfm.RelationBlock([int(v) for v in item[p]], side)], **kw)
    for sa, sb, sc in zip(a.predictor_.samples, b.predictor_.samples, c.predictor_.samples):
        np.testing.assert_allclose(sb.V, sa.V, rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(sc.V, sa.V, rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(sb.w, sa.w, rtol=1e-7, atol=1e-9)


def test_block(myfm):
    # tests/regression/test_block.py:80-149
    main, X_flat, blocks, y, group_shapes = ds.block_design()
    rbs = [myfm.RelationBlock(list(m), b) for m, b in blocks]
    fm_flat = myfm.MyFMRegressor(2, fit_w0=False).fit(X_flat, y, group_shapes=group_shapes, n_iter=30, n_kept_samples=30)
    fm_blocked = myfm.MyFMRegressor(2, fit_w0=False).fit(main, y, rbs, group_shapes=group_shapes, n_iter=30, n_kept_samples=30)
    for a, b in zip(fm_flat.predictor_.samples, fm_blocked.predictor_.samples):
        np.testing.assert_allclose(a.V, b.V)
    with tempfile.TemporaryFile() as fs:
        pickle.dump(fm_blocked, fs)
        del fm_blocked
        fs.seek(0)
        fm_blocked = pickle.load(fs)
    p1 = fm_flat.predict(main, rbs, n_workers=2)
    p2 = fm_blocked.predict(X_flat, n_workers=None)
    np.testing.assert_allclose(p1, p2)
    # doc/source/relation-blocks.rst:205-210
    assert np.abs(fm_flat.w_samples[:3] - fm_blocked.w_samples[:3]).max() < 1e-5


def test_block_multihot_matches_oracle(myfm, oracle):
    main, X_flat, blocks, y, shapes = ds.multihot_block_design()
    rbs = [myfm.RelationBlock(list(m), b) for m, b in blocks]
    fm = myfm.MyFMRegressor(3).fit(main, y, rbs, group_shapes=shapes, n_iter=12, n_kept_samples=12)
    samples, _, _ = oracle.fit(main, y, blocks, rank=3, group_index=ds.group_index_from_shapes(shapes), n_iter=12,
                               n_kept_samples=12)
    for got, want in zip(fm.predictor_.samples, samples):
        np.testing.assert_allclose(got.V, want[2], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(got.w, want[1], rtol=1e-6, atol=1e-7)
    # X=None with blocks only (base.py:230-233)
    fm2 = myfm.MyFMRegressor(2).fit(None, y, rbs, n_iter=5)
    assert fm2.predict(None, rbs).shape == y.shape


@pytest.mark.parametrize("alpha_inv", [0.3, 1.0, 3])
def test_middle_reg(myfm, alpha_inv):
    # tests/regression/test_fit.py:19-72
    X, score = ds.middle_data()
    y = score + alpha_inv * np.random.RandomState(0).normal(0, 1, size=score.shape)
    cb = RunningMean(X)
    init = myfm.MyFMGibbsRegressor(3)
    assert init.w0_samples is None and init.w_samples is None and init.V_samples is None
    fm = init.fit(X, y, X_test=X, y_test=y, n_iter=100, n_kept_samples=100, callback=cb)
    np.testing.assert_allclose(fm.predict(X), cb.sum / 100)
    last_alpha = fm.get_hyper_trace()["alpha"].iloc[-20:].values
    assert np.all(last_alpha > (1 / alpha_inv ** 2) / 2) and np.all(last_alpha < (1 / alpha_inv ** 2) * 2)
    assert np.all(np.abs(fm.w0_samples[-20:] - ds.STUB_W0) < 0.5)
    for w_ in fm.w_samples[-20:]:
        assert np.all(np.abs(w_ - ds.STUB_W) < 1.0)
    for V_ in fm.V_samples[-20:]:
        for i in range(3):
            for j in range(i + 1, 3):
                cross = ds.STUB_V[:, i].dot(ds.STUB_V[:, j])
                if abs(cross) < 0.1:
                    continue
                sign = cross / abs(cross)
                c = V_[i].dot(V_[j])
                assert sign * cross * 0.5 < c < sign * cross * 2


def test_middle_clf(myfm):
    # tests/classification/test_classification.py:13-70
    from myfm_amd.estimators import std_cdf

    X, score = ds.middle_data()
    sn = score + np.random.RandomState(0).normal(0, 1, size=score.shape)
    sn -= sn.mean()
    y = sn > 0
    cb = RunningMean(X, std_cdf)
    fm = myfm.MyFMGibbsClassifier(3).fit(X, y, X_test=X, y_test=y, n_iter=200, n_kept_samples=200, callback=cb)
    np.testing.assert_allclose(fm.predict_proba(X), cb.sum / 200)
    assert ((fm.predict(X) == y).mean()) > 0.85
    for s in fm.predictor_.samples[-20:]:
        for i in range(3):
            for j in range(i + 1, 3):
                cross = ds.STUB_V[:, i].dot(ds.STUB_V[:, j])
                if abs(cross) < 0.5:
                    continue
                sign = cross / abs(cross)
                c = np.asarray(s.V)[i].dot(np.asarray(s.V)[j])
                assert sign * cross * 0.5 < c < sign * cross * 2


def test_classification_posterior_mean_vs_oracle(myfm, oracle):
    # distributional parity: the latent z are drawn from a different (Philox) stream than the
    # reference's mt19937, so compare posterior means of P(y=1) with a Monte-Carlo tolerance
    from myfm_amd.estimators import std_cdf

    X, score = ds.middle_data()
    sn = score + np.random.RandomState(1).normal(0, 1, size=score.shape)
    y = sn > np.median(sn)
    fm = myfm.MyFMGibbsClassifier(3).fit(X, y, n_iter=400, n_kept_samples=300)
    samples, _, _ = oracle.fit(X, y.astype(float) * 2 - 1, rank=3, n_iter=400, n_kept_samples=300, task=oracle.CLASSIFICATION)
    want = np.mean([std_cdf(oracle.OracleDesign(X).predict_score(*s)) for s in samples], axis=0)
    got = fm.predict_proba(X)
    assert np.sqrt(np.mean((got - want) ** 2)) < 0.03


def test_oprobit(myfm):
    # tests/oprobit/test_oprobit_1dim.py:10-61
    from myfm_amd.estimators import std_cdf

    n = 1000
    cps = np.array([0.0, 0.5, 1.5])
    rns = np.random.RandomState(0)
    x = rns.normal(0, 2, size=n)
    y = np.zeros(n)
    score = x * 0.5 + rns.randn(n)
    for c in cps:
        y += (score > c).astype(np.int64)
    fm = myfm.MyFMOrderedProbit(0, fit_w0=False)
    fm.fit(x[:, None], y, n_iter=100, n_kept_samples=100)
    for c1, c2, c3 in fm.cutpoint_samples[-10:]:
        assert abs(c1) < 0.25 and abs(c2 - c1 - 0.5) < 0.25 and abs(c3 - c1 - 1.5) < 0.25
    p = fm.predict_proba(x[:, None])
    manual = np.zeros((n, 4))
    for s in fm.predictor_.samples:
        sc = s.predict_score(x[:, None], [])
        cdf = std_cdf(s.cutpoints[0][None, :] - sc[:, None])
        d = np.hstack([np.zeros((n, 1)), cdf, np.ones((n, 1))])
        manual += d[:, 1:] - d[:, :-1]
    np.testing.assert_allclose(manual / len(fm.predictor_.samples), p)
    assert (fm.predict(x[:, None]) == y).mean() > 0.4
    assert fm.history_.n_mh_accept[0] > 20


def test_oprobit_eval_matches_oracle_cutpoints(myfm, oracle):
    # the device likelihood/gradient/Hessian drive the same Newton start point as the oracle's
    n = 2000
    rns = np.random.RandomState(3)
    x = rns.normal(0, 1.5, size=n)
    score = 0.8 * x + rns.randn(n)
    y = (score > -0.5).astype(float) + (score > 0.4) + (score > 1.1)
    t = oracle.OracleTrainer(sps.csr_matrix(x[:, None]), y, rank=0, fit_w0=False, task=oracle.ORDERED)
    fm = myfm.MyFMOrderedProbit(0, fit_w0=False).fit(x[:, None], y, n_iter=1, n_kept_samples=1)
    # after iteration 1 both chains hold cutpoints close to the same posterior mode
    t.step()
    np.testing.assert_allclose(fm.cutpoint_samples[0], t.cutpoints(0), atol=0.15)


def test_error_behaviour(myfm):
    X, y = ds.toy()
    with pytest.raises(RuntimeError, match="index mapping points to non-existing row"):
        myfm.RelationBlock([0, 5], sps.csr_matrix(np.eye(2)))
    fm = myfm.MyFMRegressor(2).fit(X, y, n_iter=6)
    with pytest.raises(ValueError, match="Told to predict for"):
        fm.predict(X[:, :5])
    with pytest.raises(ValueError, match="Total feature size mismatch"):
        fm.predictor_.samples[0].predict_score(X[:, :5], [])
    with pytest.raises(RuntimeError, match="Shape mismatch"):
        from myfm_amd import _myfm

        cfg = _myfm.ConfigBuilder().set_identical_groups(9).build()
        _myfm.create_train_fm(2, 0.1, X, [], np.zeros(3), 1, cfg, lambda *a: False)
    # early stop through the callback (FMTrainer.hpp:78-81)
    fm = myfm.MyFMRegressor(2).fit(X, y, n_iter=50, n_kept_samples=50, callback=lambda i, *a: (i == 9, None))
    assert len(fm.history_.hypers) == 10 and len(fm.predictor_.samples) == 10
