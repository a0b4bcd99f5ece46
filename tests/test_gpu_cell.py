"""The cell path of update_V (myfm_amd/csrc/mfm_cell.hpp: designs whose rows are index tuples -- one-hot main fields + relation
blocks; no q-cache in HBM, one streaming pass per field and factor) against the CPU oracle's chain and against the generic
relation-block path of the same library, through the C ABI.  FMTrainer.hpp:315-482."""
import numpy as np
import pytest

from . import datasets as ds
from .gibbs_driver import CapiGibbs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from myfm_amd import _capi

    if _capi.lib().mfm_device_count() < 1:
        pytest.fail("no HIP device visible: the gpu tests need a real MI355X")
    return _capi


SHAPES = {
    # user + item fields, user / item / two context blocks: streams U, I (16-bit), C, C -- the shape of BASELINE configs[4]
    "u_i_ctx": dict(n_rows=60000, n_users=3000, n_items=5000, ctx=(50, 37)),
    # the item index needs 32 bits
    "items32": dict(n_rows=150000, n_users=2000, n_items=70000, ctx=(50,), item_cols=12),
    # no large scattered stream at all: user field + user block + context blocks
    "no_item": dict(n_rows=40000, n_users=2500, n_items=10, ctx=(64, 33), with_item_field=False, with_item_block=False),
    # three main fields (the third small), item block but no item-side context
    "three_fields": dict(n_rows=50000, n_users=1500, n_items=4500, ctx=(), third_field=24),
    # few items: every stream but U is an LDS table; the user block has fewer rows than the user field has columns
    # three one-hot fields and NO relation block (the flat form; the comparison context runs the multi-level fused tile pass)
    "flat_three_fields": dict(n_rows=50000, n_users=1500, n_items=4500, ctx=(), third_field=24, with_user_block=False,
                              with_item_block=False),
    "small_items": dict(n_rows=30000, n_users=1200, n_items=300, ctx=(40,), user_max=1000, user_block_rows=1000),
}


def _pair(oracle, capi, X, y, gi, rank, blocks):
    t = oracle.OracleTrainer(X, y, blocks, rank=rank, group_index=gi)
    c = capi.Context(X, y, blocks, rank=rank, group_index=gi)
    c.set_state(*t.fm())
    c.set_e(t.e(X.shape[0]))
    return t, c


@pytest.mark.parametrize("groups", ["5", "23"])
@pytest.mark.parametrize("shape", list(SHAPES))
def test_cell_chain_matches_oracle_and_generic_path(oracle, capi, monkeypatch, shape, groups):
    monkeypatch.setenv("MFM_CELL_MIN_ROWS", "0")
    monkeypatch.setenv("MFM_CELL_GROUPS", groups)
    main, blocks, y, shapes = ds.tuple_design(**SHAPES[shape])
    gi = ds.group_index_from_shapes(shapes)
    rank, n = 3, main.shape[0]
    t, c = _pair(oracle, capi, main, y, gi, rank, blocks)
    assert c.plan_flags()["cell"]
    t0 = t.clone()
    d = CapiGibbs(c, t.clone(), n, gi)
    monkeypatch.setenv("MFM_NO_CELL", "1")
    _, cg = _pair(oracle, capi, main, y, gi, rank, blocks)
    monkeypatch.delenv("MFM_NO_CELL")
    assert not cg.plan_flags()["cell"]
    dg = CapiGibbs(cg, t.clone(), n, gi)
    for it in range(3):
        t.step()
        d.step()
        dg.step()
        w0, w, V = t.fm()
        _, gw, gV = c.get_state()
        _, hw, hV = cg.get_state()
        np.testing.assert_allclose(gV, V, rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(gw, w, rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(gV, hV, rtol=1e-7, atol=1e-8)  # cell path == generic relation-block path
        assert abs(d.alpha - t.hyper()["alpha"]) < 1e-7 * d.alpha
    np.testing.assert_allclose(c.get_e(), t.e(n), rtol=1e-7, atol=1e-7)
    # every sum has a fixed order: a second context from the same state is bit-identical
    _, c2 = _pair(oracle, capi, main, y, gi, rank, blocks)
    c2.set_state(*t0.fm())
    c2.set_e(t0.e(n))
    d2 = CapiGibbs(c2, t0.clone(), n, gi)
    for it in range(3):
        d2.step()
    assert np.array_equal(c2.get_state()[2], c.get_state()[2]) and np.array_equal(c2.get_e(), c.get_e())


def test_cell_single_sweep_residual_and_q(oracle, capi, monkeypatch):
    # one update_V call on the cell path: coefficients, the residual it leaves in row order and the q-cache it owes
    # (q_train as the reference leaves it, FMTrainer.hpp:373 / :479) against the oracle
    monkeypatch.setenv("MFM_CELL_MIN_ROWS", "0")
    monkeypatch.setenv("MFM_CELL_GROUPS", "7")
    main, blocks, y, shapes = ds.tuple_design(**SHAPES["u_i_ctx"])
    gi = ds.group_index_from_shapes(shapes)
    rank, n = 2, main.shape[0]
    t, c = _pair(oracle, capi, main, y, gi, rank, blocks)
    assert c.plan_flags()["cell"]
    G, D = t.G, t.D
    rng = np.random.default_rng(5)
    lam = rng.uniform(0.5, 2.0, size=(G, rank))
    mu = rng.normal(size=(G, rank)) * 0.1
    h = t.hyper()
    t.set_hyper(0.7, h["mu_w"], h["lambda_w"], mu, lam)
    z = t.clone().rng_sample_normals(rank * D)
    for f in range(rank):
        t.update_V_factor(f)
    c.sweep_V(0, rank, 0.7, lam, mu, z)
    np.testing.assert_allclose(c.get_state()[2], t.fm()[2], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(c.get_e(), t.e(n), rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(c.get_q(), t.q(n), rtol=1e-8, atol=1e-9)


@pytest.mark.parametrize("shape", ["u_i_ctx", "items32", "no_item", "three_fields"])
@pytest.mark.parametrize("rank", [3, 10])
def test_cell_scorer(oracle, capi, monkeypatch, shape, rank):
    # update_e on the cell layout (K / FB passes with the factor tables in LDS) against the oracle's FM::predict_score
    # (FM.hpp:54-136) and against the library's generic one-pass scorer
    monkeypatch.setenv("MFM_CELL_MIN_ROWS", "0")
    monkeypatch.setenv("MFM_CELL_GROUPS", "6")
    main, blocks, y, shapes = ds.tuple_design(**SHAPES[shape])
    gi = ds.group_index_from_shapes(shapes)
    n = main.shape[0]
    t, c = _pair(oracle, capi, main, y, gi, rank, blocks)
    assert c.plan_flags()["cell"]
    t.substep(8)
    c.update_e_regression()
    e = c.get_e()
    np.testing.assert_allclose(e, t.e(n), rtol=1e-10, atol=1e-10)
    monkeypatch.setenv("MFM_NO_CELL", "1")
    _, cg = _pair(oracle, capi, main, y, gi, rank, blocks)
    cg.update_e_regression()
    np.testing.assert_allclose(e, cg.get_e(), rtol=1e-11, atol=1e-11)


@pytest.mark.parametrize("shape", list(SHAPES))
def test_cell_linear_sweep(oracle, capi, monkeypatch, shape):
    # update_w on the cell layout (one sum per index value, no q tables; FMTrainer.hpp:231-313): coefficients and the residual
    # after ONE call against the oracle, and the generic sweep (MFM_NO_CELL_W) of the same library
    monkeypatch.setenv("MFM_CELL_MIN_ROWS", "0")
    monkeypatch.setenv("MFM_CELL_GROUPS", "9")
    main, blocks, y, shapes = ds.tuple_design(**SHAPES[shape])
    gi = ds.group_index_from_shapes(shapes)
    n = main.shape[0]
    t, c = _pair(oracle, capi, main, y, gi, 2, blocks)
    assert c.plan_flags()["cell"]
    G, D = t.G, t.D
    rng = np.random.default_rng(6)
    lam, mu = rng.uniform(0.5, 2.0, size=G), rng.normal(size=G) * 0.1
    h = t.hyper()
    t.set_hyper(1.3, mu, lam, h["mu_V"], h["lambda_V"])
    z = t.clone().rng_sample_normals(D)
    t.substep(4)
    c.sweep_w(1.3, lam, mu, z)
    np.testing.assert_allclose(c.get_state()[1], t.fm()[1], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(c.get_e(), t.e(n), rtol=1e-9, atol=1e-10)
    monkeypatch.setenv("MFM_NO_CELL_W", "1")
    t2, cg = _pair(oracle, capi, main, y, gi, 2, blocks)
    cg.sweep_w(1.3, lam, mu, z)
    np.testing.assert_allclose(c.get_state()[1], cg.get_state()[1], rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("task", ["regression", "classification", "ordered"])
def test_cell_path_through_the_trainer(oracle, monkeypatch, task):
    # create_train_fm (the drop-in boundary) on an index-tuple design with the cell path forced: every kept sample and
    # hyper-parameter draw of the three tasks against the oracle's chain. Classification / ordered probit in the host-RNG test
    # mode (tests/test_gpu_host_rng_parity.py), so that the latent draws are the reference's, row by row: the cell path's
    # update_w / update_V / update_e feed kernels that work on eq in row order (the residual is brought back on demand).
    from myfm_amd import _myfm

    from .test_gpu_baseline_configs import _assert_chain, _config
    from .test_gpu_host_rng_parity import _oracle_chain

    monkeypatch.setenv("MFM_CELL_MIN_ROWS", "0")
    monkeypatch.setenv("MFM_CELL_GROUPS", "11")
    if task != "regression":
        monkeypatch.setenv("MYFM_AMD_HOST_RNG", "1")
    main, blocks, score, shapes = ds.tuple_design(n_rows=25000, n_users=900, n_items=4500, ctx=(30,))
    n = main.shape[0]
    gi = ds.group_index_from_shapes(shapes)
    n_iter, rank = 4, 3
    kw, groups = {}, None
    if task == "regression":
        y = score
    elif task == "classification":
        y = np.where(score > np.median(score), 1.0, -1.0)
        kw = dict(task=oracle.CLASSIFICATION)
    else:
        s = (score - score.mean()) / score.std()
        y = np.zeros(n)
        for c in (-0.8, 0.0, 0.9):
            y += s > c
        groups = [(4, np.arange(n))]
        kw = dict(task=oracle.ORDERED, cutpoint_groups=groups)
    rels = [_myfm.RelationBlock(np.asarray(m, dtype=np.int64), B) for m, B in blocks]
    cfg = _config(_myfm, gi, n_iter, n_iter, task=task, **({"cutpoint_groups": groups} if groups else {}))
    predictor, history = _myfm.create_train_fm(rank, 0.1, main, rels, y, 42, cfg, lambda *a: False)
    samples, hypers, cuts, t = _oracle_chain(oracle, main, y, blocks, n_iter, n_groups_cut=1 if groups else 0, rank=rank,
                                             group_index=gi, **kw)
    _assert_chain(predictor, history, samples, hypers)
    if groups:
        for fm, cut in zip(predictor.samples, cuts):
            np.testing.assert_allclose(fm.cutpoints[0], cut[0], rtol=1e-7, atol=1e-7)
        assert list(history.n_mh_accept) == [t.mh_accept(0)]
