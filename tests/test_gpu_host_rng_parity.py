"""Draw-for-draw parity of the probit-classification and ordered-probit chains (FMTrainer.hpp:498-521,
OProbitSampler.hpp:238-272, :359-387, util.hpp:15-78).

The latent draws of those tasks consume the reference's generator in data-dependent rejection loops, row after row; the
product draws them on the device from per-row Philox streams (parity distributional, tests/test_gpu_task_kernels.py). Under
MYFM_AMD_HOST_RNG=1 -- a TEST mode -- the trainer keeps its std::mt19937 on the host and makes those draws there, in the
reference's row order, from the device's scores, so that the whole chain (sweeps, scorer, cutpoint likelihood on the
device) can be held against the CPU oracle like the regression chains: 1e-7 per kept sample and hyper-parameter draw.
"""
import numpy as np
import pytest
import scipy.sparse as sps

from . import datasets as ds
from .test_gpu_baseline_configs import _assert_chain, _config

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from myfm_amd import _capi, _myfm

    if _myfm.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu tests need a real MI355X")
    return _myfm, _capi


def _oracle_chain(oracle, X, y, blocks, n_iter, n_groups_cut=0, **kw):
    t = oracle.OracleTrainer(X, y, blocks, **kw)
    samples, hypers, cuts = [], [], []
    for it in range(n_iter):
        t.step()
        samples.append(t.fm())
        hypers.append(t.hyper())
        cuts.append([t.cutpoints(g) for g in range(n_groups_cut)])
    return samples, hypers, cuts, t


def _blocks_design(seed=0, n=6000):
    """main table + two relation blocks (the shape of tests/regression/test_block.py:81-113, larger)"""
    rns = np.random.RandomState(seed)
    n_u, n_i = 300, 120
    u = np.sort(rns.randint(0, n_u, size=n))
    i = rns.randint(0, n_i, size=n)
    main = sps.csr_matrix(rns.normal(size=(n, 1)))
    Bu = sps.hstack([sps.identity(n_u), sps.csr_matrix(rns.normal(size=(n_u, 2)))]).tocsr()
    Bi = sps.hstack([sps.identity(n_i), sps.csr_matrix((rns.rand(n_i, 3) > 0.5).astype(np.float64))]).tocsr()
    shapes = [1, n_u, 2, n_i, 3]
    score = 0.4 * main.toarray()[:, 0] + rns.normal(size=n_u)[u] * 0.8 + rns.normal(size=n_i)[i] * 0.5 + rns.normal(size=n) * 0.5
    return main, [(u, Bu), (i, Bi)], score, shapes


@pytest.mark.parametrize("design", ["onehot", "blocks"])
def test_classification_chain_draw_for_draw(mods, oracle, monkeypatch, design):
    _myfm, _ = mods
    monkeypatch.setenv("MYFM_AMD_HOST_RNG", "1")
    if design == "onehot":
        n = 30000
        X, score, shapes = ds.onehot_mf(n, 400, 150, seed=4, sort_by_user=True)
        blocks = []
        score = score - np.median(score)
    else:
        X, blocks, score, shapes = _blocks_design()
    y = np.where(score > 0, 1.0, -1.0)  # (base.py:385-386: targets mapped to +-1)
    gi = ds.group_index_from_shapes(shapes)
    n_iter, rank = 5, 4
    rels = [_myfm.RelationBlock(np.asarray(m, dtype=np.int64), B) for m, B in blocks]
    cfg = _config(_myfm, gi, n_iter, n_iter, task="classification")
    predictor, history = _myfm.create_train_fm(rank, 0.1, X, rels, y, 42, cfg, lambda *a: False)
    samples, hypers, _, _ = _oracle_chain(oracle, X, y, blocks, n_iter, rank=rank, group_index=gi, task=oracle.CLASSIFICATION)
    _assert_chain(predictor, history, samples, hypers)


@pytest.mark.parametrize("design", ["one_group", "two_groups", "blocks"])
def test_ordered_probit_chain_draw_for_draw(mods, oracle, monkeypatch, design):
    _myfm, _ = mods
    monkeypatch.setenv("MYFM_AMD_HOST_RNG", "1")
    if design == "blocks":
        X, blocks, score, shapes = _blocks_design(seed=2)
        n = X.shape[0]
    else:
        n = 20000
        X, score, shapes = ds.onehot_mf(n, 300, 100, seed=9, sort_by_user=True)
        blocks = []
    score = (score - score.mean()) / score.std()
    if design == "two_groups":
        rows_a, rows_b = np.arange(0, n, 2), np.arange(1, n, 2)
        y = np.zeros(n)
        for c in (-0.4, 0.5):
            y[rows_a] += score[rows_a] > c
        for c in (-0.8, 0.0, 0.9):
            y[rows_b] += score[rows_b] > c
        groups = [(3, rows_a), (4, rows_b)]
    else:
        y = np.zeros(n)
        for c in (-0.9, -0.2, 0.4, 1.1):
            y += score > c
        groups = [(5, np.arange(n))]
    gi = ds.group_index_from_shapes(shapes)
    n_iter, rank = 5, 3
    rels = [_myfm.RelationBlock(np.asarray(m, dtype=np.int64), B) for m, B in blocks]
    cfg = _config(_myfm, gi, n_iter, n_iter, task="ordered", cutpoint_groups=groups)
    predictor, history = _myfm.create_train_fm(rank, 0.1, X, rels, y, 42, cfg, lambda *a: False)
    samples, hypers, cuts, t = _oracle_chain(oracle, X, y, blocks, n_iter, n_groups_cut=len(groups), rank=rank, group_index=gi,
                                             task=oracle.ORDERED, cutpoint_groups=groups)
    _assert_chain(predictor, history, samples, hypers)
    for fm, cut in zip(predictor.samples, cuts):
        for g in range(len(groups)):
            np.testing.assert_allclose(fm.cutpoints[g], cut[g], rtol=1e-7, atol=1e-7)
    assert list(history.n_mh_accept) == [t.mh_accept(g) for g in range(len(groups))]


def test_exact_latent_draws_is_a_supported_option(mods, oracle, monkeypatch):
    """The same mode without the environment switch: `MyFMClassifier(..., exact_latent_draws=True)` (the estimator keeps the
    caller's row order and sets `ConfigBuilder.set_exact_latent_draws`) reproduces the oracle's probit chain draw for draw on
    rows that arrive UNSORTED; the opt-out (per-row Philox streams) gives another chain of the same law."""
    from myfm_amd import MyFMClassifier

    monkeypatch.delenv("MYFM_AMD_HOST_RNG", raising=False)
    n = 20000
    X, score, shapes = ds.onehot_mf(n, 300, 100, seed=11, sort_by_user=False)
    y = (score - np.median(score)) > 0
    n_iter, rank = 4, 3
    fm = MyFMClassifier(rank, random_seed=7, exact_latent_draws=True)
    fm.fit(X, y, n_iter=n_iter, n_kept_samples=n_iter, group_shapes=shapes)
    gi = ds.group_index_from_shapes(shapes)
    samples, hypers, _, _ = _oracle_chain(oracle, X, np.where(y, 1.0, -1.0), [], n_iter, rank=rank, group_index=gi,
                                          task=oracle.CLASSIFICATION, seed=7)
    _assert_chain(fm.predictor_, fm.history_, samples, hypers)
    fm2 = MyFMClassifier(rank, random_seed=7, exact_latent_draws=False)
    fm2.fit(X, y, n_iter=n_iter, n_kept_samples=n_iter, group_shapes=shapes)
    assert not np.allclose(fm2.V_samples[-1], fm.V_samples[-1], rtol=1e-7, atol=1e-7)
    p1, p2 = fm.predict_proba(X[:2000]), fm2.predict_proba(X[:2000])
    assert np.sqrt(np.mean((p1 - p2) ** 2)) < 0.1
