"""Latent mode "exact": the latent draws of probit classification / ordered probit (FMTrainer.hpp:498-521,
OProbitSampler.hpp:238-272, util.hpp:15-60) made ON THE DEVICE from the reference's own random stream (csrc/mfm_latent.hip:
coalescing flows over (row, quad)), the cutpoint sampler's Metropolis draws (OProbitSampler.hpp:55-72, :378) served from the same
device stream. Every chain below is held against the CPU oracle draw for draw: kept samples, hyper-parameters, cutpoints,
Metropolis accept counts at 1e-7 -- the same bar as the regression chains.
"""
import numpy as np
import pytest

from . import datasets as ds
from .test_gpu_baseline_configs import _assert_chain
from .test_gpu_host_rng_parity import _blocks_design, _oracle_chain

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from myfm_amd import _capi, _myfm

    if _myfm.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu tests need a real MI355X")
    return _myfm, _capi


def _config(_myfm, gi, n_iter, task, cutpoint_groups=None, latent="exact"):
    b = _myfm.ConfigBuilder()
    b.set_alpha_0(1.0).set_beta_0(1.0).set_gamma_0(1.0).set_mu_0(0.0).set_reg_0(1.0)
    b.set_group_index([int(g) for g in gi]).set_n_iter(n_iter).set_n_kept_samples(n_iter)
    b.set_task_type({"classification": _myfm.TaskType.CLASSIFICATION, "ordered": _myfm.TaskType.ORDERED}[task])
    if cutpoint_groups is not None:
        b.set_cutpoint_groups([(int(c), [int(r) for r in rows]) for c, rows in cutpoint_groups])
    b.set_latent_mode(latent)
    return b.build()


def _run_session(_myfm, rank, X, rels, y, cfg, n_iter):
    """the same loop as create_train_fm, steppable: returns (samples, hypers, cutpoints per iteration, session)"""
    sess = _myfm.GibbsSession(rank, 0.1, X, rels, y, 42, cfg)
    samples, hypers, cuts = [], [], []
    for _ in range(n_iter):
        sess.step()
        fm = sess.fm
        samples.append((fm.w0, np.array(fm.w), np.array(fm.V)))
        h = sess.hyper  # (a reference to the session's live object: copy the values)
        hypers.append(dict(lambda_w=np.array(h.lambda_w), mu_w=np.array(h.mu_w), lambda_V=np.array(h.lambda_V), mu_V=np.array(h.mu_V)))
        cuts.append([np.array(c) for c in fm.cutpoints])
    return samples, hypers, cuts, sess


def _assert_session(samples, hypers, want_samples, want_hypers, tol=1e-7):
    for (w0, w, V), (w0o, wo, Vo) in zip(samples, want_samples):
        assert abs(w0 - w0o) <= tol * max(1.0, abs(w0o))
        np.testing.assert_allclose(w, wo, rtol=tol, atol=tol)
        np.testing.assert_allclose(V, Vo, rtol=tol, atol=tol)
    for hd, ho in zip(hypers, want_hypers):
        for k in ("lambda_w", "lambda_V"):
            np.testing.assert_allclose(hd[k], ho[k], rtol=tol)
        for k in ("mu_w", "mu_V"):
            np.testing.assert_allclose(hd[k], ho[k], rtol=tol, atol=tol)


def _classification_case(design):
    if design == "onehot":
        X, score, shapes = ds.onehot_mf(30000, 400, 150, seed=4, sort_by_user=True)
        blocks = []
        score = score - np.median(score)
    else:
        X, blocks, score, shapes = _blocks_design()
    return X, blocks, np.where(score > 0, 1.0, -1.0), shapes


def _ordered_case(design):
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
    return X, blocks, y, shapes, groups


@pytest.mark.parametrize("design", ["onehot", "blocks"])
def test_classification_chain_exact_on_device(mods, oracle, design):
    _myfm, _ = mods
    X, blocks, y, shapes = _classification_case(design)
    gi = ds.group_index_from_shapes(shapes)
    n_iter, rank = 5, 4
    rels = [_myfm.RelationBlock(np.asarray(m, dtype=np.int64), B) for m, B in blocks]
    predictor, history = _myfm.create_train_fm(rank, 0.1, X, rels, y, 42, _config(_myfm, gi, n_iter, "classification"), lambda *a: False)
    samples, hypers, _, _ = _oracle_chain(oracle, X, y, blocks, n_iter, rank=rank, group_index=gi, task=oracle.CLASSIFICATION)
    _assert_chain(predictor, history, samples, hypers)
    # ... and it was the parallel evaluation that made the draws, not its sequential fall-back
    got, gh, _, sess = _run_session(_myfm, rank, X, rels, y, _config(_myfm, gi, n_iter, "classification"), n_iter)
    _assert_session(got, gh, samples, hypers)
    info = sess.latent_info()
    assert info["mode"] == "exact" and info["sequential_fallbacks"] == 0 and info["status"] == 0, info
    assert info["quads_consumed"] >= X.shape[0]


@pytest.mark.parametrize("design", ["one_group", "two_groups", "blocks"])
def test_ordered_probit_chain_exact_on_device(mods, oracle, design):
    _myfm, _ = mods
    X, blocks, y, shapes, groups = _ordered_case(design)
    gi = ds.group_index_from_shapes(shapes)
    n_iter, rank = 5, 3
    rels = [_myfm.RelationBlock(np.asarray(m, dtype=np.int64), B) for m, B in blocks]
    cfg = _config(_myfm, gi, n_iter, "ordered", cutpoint_groups=groups)
    predictor, history = _myfm.create_train_fm(rank, 0.1, X, rels, y, 42, cfg, lambda *a: False)
    samples, hypers, cuts, t = _oracle_chain(oracle, X, y, blocks, n_iter, n_groups_cut=len(groups), rank=rank, group_index=gi,
                                             task=oracle.ORDERED, cutpoint_groups=groups)
    _assert_chain(predictor, history, samples, hypers)
    for fm, cut in zip(predictor.samples, cuts):
        for g in range(len(groups)):
            np.testing.assert_allclose(fm.cutpoints[g], cut[g], rtol=1e-7, atol=1e-7)
    assert list(history.n_mh_accept) == [t.mh_accept(g) for g in range(len(groups))]
    _, _, _, sess = _run_session(_myfm, rank, X, rels, y, cfg, 2)
    info = sess.latent_info()
    assert info["mode"] == "exact" and info["sequential_fallbacks"] == 0, info


@pytest.mark.parametrize("env", [
    {"MFM_LAT_CHUNKS": "64", "MFM_LAT_MIN_LQ": "256", "MFM_LAT_SUBQ": "64", "MFM_LAT_ROUND": "4"},      # many short chunks
    {"MFM_LAT_CHUNKS": "7", "MFM_LAT_MIN_LQ": "1024", "MFM_LAT_SUBQ": "128", "MFM_LAT_NO_RESIDENT": "1"},  # launches only
    {"MFM_LAT_CHUNKS": "1"},                                                                               # one chunk: one walker
    {"MFM_LAT_CHUNKS": "200", "MFM_LAT_MIN_LQ": "128", "MFM_LAT_SUBQ": "128", "MFM_LAT_ROUND": "128"},
])
def test_every_geometry_gives_the_same_draws(mods, oracle, monkeypatch, env):
    """chunk length, sub-chunk length, round length and the hand-over to the resident kernel only change HOW the one sequential
    path is found: the chain is the oracle's for every choice"""
    _myfm, _ = mods
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    X, blocks, y, shapes, groups = _ordered_case("one_group")
    gi = ds.group_index_from_shapes(shapes)
    n_iter, rank = 3, 3
    cfg = _config(_myfm, gi, n_iter, "ordered", cutpoint_groups=groups)
    got, gh, gc, sess = _run_session(_myfm, rank, X, [], y, cfg, n_iter)
    samples, hypers, cuts, _ = _oracle_chain(oracle, X, y, [], n_iter, n_groups_cut=1, rank=rank, group_index=gi, task=oracle.ORDERED,
                                             cutpoint_groups=groups)
    _assert_session(got, gh, samples, hypers)
    for a, b in zip(gc, cuts):
        np.testing.assert_allclose(a[0], b[0], rtol=1e-7, atol=1e-7)
    info = sess.latent_info()
    assert info["sequential_fallbacks"] == 0, info


@pytest.mark.parametrize("retry", ["0.02", "6.5"])
def test_a_missed_window_falls_back_to_the_same_draws(mods, oracle, monkeypatch, retry):
    """windows of +-0.02 sigma cannot hold the path: the parallel evaluation reports it without having drawn or consumed anything.
    retry 6.5: the second attempt with windows of +-6.5 sigma holds it (what happens to one draw in ~100 with the default +-3.5 sigma);
    retry 0.02: the second attempt misses too and the sequential loop makes the same draws from the same stream position (the host's
    window into the device stream)."""
    _myfm, _ = mods
    monkeypatch.setenv("MFM_LAT_KSIGMA", "0.02")
    monkeypatch.setenv("MFM_LAT_KSIGMA_RETRY", retry)
    monkeypatch.setenv("MFM_LAT_CHUNKS", "16")
    monkeypatch.setenv("MFM_LAT_MIN_LQ", "512")
    for task in ("classification", "ordered"):
        if task == "classification":
            X, blocks, y, shapes = _classification_case("onehot")
            groups = None
        else:
            X, blocks, y, shapes, groups = _ordered_case("one_group")
        gi = ds.group_index_from_shapes(shapes)
        n_iter, rank = 3, 3
        cfg = _config(_myfm, gi, n_iter, task, cutpoint_groups=groups)
        got, gh, gc, sess = _run_session(_myfm, rank, X, [], y, cfg, n_iter)
        samples, hypers, cuts, _ = _oracle_chain(oracle, X, y, [], n_iter, n_groups_cut=1 if groups else 0, rank=rank, group_index=gi,
                                                 task=oracle.ORDERED if groups else oracle.CLASSIFICATION, cutpoint_groups=groups)
        _assert_session(got, gh, samples, hypers)
        info = sess.latent_info()
        if retry == "0.02":
            assert info["sequential_fallbacks"] >= n_iter, info
        else:
            assert info["sequential_fallbacks"] == 0 and info["attempts"] == 2 and info["status"] == 0, info


@pytest.mark.parametrize("task", ["ordered", "classification"])
def test_exact_equals_host_mode_on_two_million_rows(mods, task):
    """2 M rows, 5 classes / probit classification: the device evaluation against the host loop over the same scores (latent mode
    "host" keeps the whole generator on the host): residual after every iteration to 1e-9, hyper-parameters, cutpoints. Classification
    at this size runs with the windows' control variate (csrc/mfm_latent.hip: shifted, narrower first-attempt windows)."""
    _myfm, _ = mods
    from myfm_amd.utils import synthetic as syn

    X, y, shapes = syn.movielens_like(2_000_000, 20000, 3000, rank_true=8, seed=3)
    if task == "ordered":
        yo = (np.clip(np.round(y), 1, 5) - 1).astype(np.float64)
    else:
        yo = np.where(y > np.median(y), 1.0, -1.0)
    gi = syn.group_index_from_shapes(shapes)
    n_iter, rank = 3, 4

    def cfg(latent):
        b = _myfm.ConfigBuilder()
        b.set_alpha_0(1.0).set_beta_0(1.0).set_gamma_0(1.0).set_mu_0(0.0).set_reg_0(1.0)
        b.set_group_index([int(g) for g in gi]).set_n_iter(n_iter).set_n_kept_samples(0)
        if task == "ordered":
            b.set_task_type(_myfm.TaskType.ORDERED).set_cutpoint_groups([(5, np.arange(X.shape[0]))])
        else:
            b.set_task_type(_myfm.TaskType.CLASSIFICATION)
        b.set_latent_mode(latent)
        return b.build()

    a = _myfm.GibbsSession(rank, 0.1, X, [], yo, 42, cfg("exact"))
    b = _myfm.GibbsSession(rank, 0.1, X, [], yo, 42, cfg("host"))
    for _ in range(n_iter):
        a.step()
        b.step()
        np.testing.assert_allclose(a.residual(), b.residual(), rtol=1e-9, atol=1e-9)
        if task == "ordered":
            np.testing.assert_allclose(np.array(a.fm.cutpoints[0]), np.array(b.fm.cutpoints[0]), rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(np.array(a.fm.V), np.array(b.fm.V), rtol=1e-8, atol=1e-8)
    info = a.latent_info()
    assert info["sequential_fallbacks"] == 0 and info["chunks"] > 100, info


@pytest.mark.parametrize("task", ["classification", "ordered"])
def test_estimators_default_is_the_reference_chain_on_unsorted_rows(mods, oracle, task):
    """MyFMClassifier / MyFMOrderedProbit as a user calls them (no switch): rows arrive UNSORTED, fit() sorts them for the device paths
    and hands the caller's order to the latent draws -- kept samples, hyper-parameters (and cutpoints) are the oracle's for the
    caller's table and seed."""
    from myfm_amd import MyFMClassifier, MyFMOrderedProbit

    n = 20000
    X, score, shapes = ds.onehot_mf(n, 300, 100, seed=11, sort_by_user=False)
    gi = ds.group_index_from_shapes(shapes)
    n_iter, rank = 4, 3
    if task == "classification":
        y = (score - np.median(score)) > 0
        fm = MyFMClassifier(rank, random_seed=7)
        fm.fit(X, y, n_iter=n_iter, n_kept_samples=n_iter, group_shapes=shapes)
        samples, hypers, _, _ = _oracle_chain(oracle, X, np.where(y, 1.0, -1.0), [], n_iter, rank=rank, group_index=gi,
                                              task=oracle.CLASSIFICATION, seed=7)
        _assert_chain(fm.predictor_, fm.history_, samples, hypers)
    else:
        sc = (score - score.mean()) / score.std()
        y = np.zeros(n)
        for c in (-0.9, -0.2, 0.4, 1.1):
            y += sc > c
        fm = MyFMOrderedProbit(rank, random_seed=7)
        fm.fit(X, y, n_iter=n_iter, n_kept_samples=n_iter, group_shapes=shapes)
        groups = [(5, np.arange(n))]
        samples, hypers, cuts, t = _oracle_chain(oracle, X, y, [], n_iter, n_groups_cut=1, rank=rank, group_index=gi, task=oracle.ORDERED,
                                                 cutpoint_groups=groups, seed=7)
        _assert_chain(fm.predictor_, fm.history_, samples, hypers)
        for f, cut in zip(fm.predictor_.samples, cuts):
            np.testing.assert_allclose(f.cutpoints[0], cut[0], rtol=1e-7, atol=1e-7)
    # the opt-out gives another chain of the same law
    est = MyFMClassifier if task == "classification" else MyFMOrderedProbit
    fm2 = est(rank, random_seed=7, exact_latent_draws=False)
    fm2.fit(X, y, n_iter=n_iter, n_kept_samples=n_iter, group_shapes=shapes)
    assert not np.allclose(fm2.V_samples[-1], fm.V_samples[-1], rtol=1e-7, atol=1e-7)


@pytest.mark.parametrize("case", ["two_classes", "seven_rows", "rank_zero", "one_class_missing", "wide_middle"])
def test_small_and_degenerate_ordered_designs(mods, oracle, case):
    """corner cases of the exact latent draws through the boundary: 2 classes (no two-sided row at all), a table of 7 rows (one
    chunk, one walker), rank 0 (tests/oprobit/test_oprobit_1dim.py:24 in the reference), a class nobody belongs to, a middle class
    that holds 98 % of the rows (cutpoints far apart: the uniform proposal is accepted one time in three, the draw consumes twice
    the engine outputs the generator was sized for -- the ring of outputs grows on the fly)"""
    _myfm, _ = mods
    rns = np.random.RandomState(3)
    if case == "seven_rows":
        n, n_u, n_i = 7, 3, 2
    else:
        n, n_u, n_i = 3000, 40, 25
    X, score, shapes = ds.onehot_mf(n, n_u, n_i, seed=21, sort_by_user=True)
    score = (score - score.mean()) / (score.std() + 1e-12) + 0.3 * rns.normal(size=n)
    if case == "two_classes":
        y = (score > 0.1).astype(np.float64)
        n_class = 2
    elif case == "wide_middle":
        y = np.ones(n)
        y[score < np.quantile(score, 0.01)] = 0.0
        y[score > np.quantile(score, 0.99)] = 2.0
        n_class = 3
    elif case == "one_class_missing":
        y = np.zeros(n)
        for c in (-0.5, 0.5):
            y += score > c
        y[y == 1] = 2.0  # classes 0 and 2 of {0, 1, 2, 3}: class 1 and class 3 are empty
        n_class = 4
    else:
        y = np.zeros(n)
        for c in (-0.6, 0.0, 0.7):
            y += score > c
        n_class = 4
    rank = 0 if case == "rank_zero" else 2
    groups = [(n_class, np.arange(n))]
    gi = ds.group_index_from_shapes(shapes)
    n_iter = 4
    cfg = _config(_myfm, gi, n_iter, "ordered", cutpoint_groups=groups)
    predictor, history = _myfm.create_train_fm(rank, 0.1, X, [], y, 42, cfg, lambda *a: False)
    samples, hypers, cuts, t = _oracle_chain(oracle, X, y, [], n_iter, n_groups_cut=1, rank=rank, group_index=gi, task=oracle.ORDERED,
                                             cutpoint_groups=groups)
    _assert_chain(predictor, history, samples, hypers)
    for fm, cut in zip(predictor.samples, cuts):
        np.testing.assert_allclose(fm.cutpoints[0], cut[0], rtol=1e-7, atol=1e-7)
    assert list(history.n_mh_accept) == [t.mh_accept(0)]


def test_c_abi_classification_draw_and_refusal(oracle):
    """The entry points themselves, through ctypes: `mfm_update_e_classification_exact` makes the draws of the oracle's `update_e`
    (FMTrainer.hpp:498-512) from the same engine state and leaves the stream where the oracle's generator stands; with a row whose
    score lies 5000 standard deviations on the wrong side of its class (alpha* > 1000: outside what the walkers' single decision
    formula covers) it returns status 5 and has neither drawn nor consumed anything -- the same call after the weights are put
    back makes the same draws."""
    from myfm_amd import _capi

    n, K = 40000, 4
    X, score, shapes = ds.onehot_mf(n, 500, 200, seed=12, sort_by_user=True)
    y = np.where(score > np.median(score), 1.0, -1.0)
    t = oracle.OracleTrainer(X, y, rank=K, seed=5, task=oracle.CLASSIFICATION)
    for _ in range(2):
        t.step()
    w0, w, V = t.fm()
    st, pos = t.rng_state()
    c = _capi.Context(X, y, rank=K)
    c.rng_seed_mt19937(st, pos)
    c.rng_set_program([(0, 0, 4, 0, 0.0), (2, 0, n, 0, 0.0)])  # (MFM_RNG_LATENT: the draws of n rows per iteration)
    ahead = c.rng_host_read(0, 8)

    # a row far out: user 0 carries y = +1 somewhere; push that user's weight to -5000
    far = w.copy()
    u0 = int(X[np.flatnonzero(y > 0)[0]].indices[0])
    far[u0] = -5000.0
    c.set_state(w0, far, V)
    assert c.update_e_classification_exact() == 5
    assert c.latent_stats()["status"] == 5
    np.testing.assert_array_equal(c.rng_host_read(0, 8), ahead)

    c.set_state(w0, w, V)
    assert c.update_e_classification_exact() == 0
    info = c.latent_stats()
    assert info["status"] == 0 and info["quads"] >= n
    e = c.get_e()
    t.substep(8)
    np.testing.assert_allclose(e, t.e(n), rtol=1e-9, atol=1e-9)
    # the stream: 4 engine outputs per quad were consumed, the next ones are the oracle generator's next ones
    st2, pos2 = t.rng_state()
    bg = np.random.MT19937()
    bg.state = {"bit_generator": "MT19937", "state": {"key": st2, "pos": pos2}}
    np.testing.assert_array_equal(c.rng_host_read(0, 8), bg.random_raw(8).astype(np.uint32))
    c.close()
