"""The streamed block-feature sweep (csrc/mfm_chain_stream.hpp, k_cs_stream): the feature chain of a relation block whose state does
not fit one CU's LDS (FMTrainer.hpp:276-302 for w, :419-470 for V) as ONE pipelined launch -- a walker wavefront over the hot rows'
records in LDS, helper wavefronts moving rows in and out, row-range workgroups taking cold statistics / applying cold updates, all
synchronised by monotonic counters. Regression chains against the CPU oracle (same seed, 1e-7), for several step widths, windows,
range counts and LDS budgets; bit-reproducible; the conflict-batched form (MFM_NO_CB_STREAM) gives the same chain."""
import numpy as np
import pytest
import scipy.sparse as sps

from . import datasets as ds
from .gibbs_driver import CapiGibbs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from myfm_amd import _capi

    if _capi.lib().mfm_device_count() < 1:
        pytest.fail("no HIP device visible: the gpu tests need a real MI355X")
    return _capi


def _multihot(rng, n_rows, n_cols, per_row, ident=False):
    cols = np.stack([rng.choice(n_cols, size=per_row, replace=False) for _ in range(n_rows)])
    vals = np.round(rng.uniform(0.2, 1.0, size=cols.shape), 3)
    B = sps.csr_matrix((vals.ravel(), cols.ravel(), np.arange(0, n_rows * per_row + 1, per_row)), shape=(n_rows, n_cols))
    B.sort_indices()
    if ident:
        B = sps.hstack([sps.identity(n_rows, format="csr"), B]).tocsr()
    return B


def _design(n=60000, seed=3, rows_a=5000, cols_a=90, per_a=6, rows_b=2600, cols_b=50, per_b=4):
    rng = np.random.default_rng(seed)
    ma, mb = rng.integers(0, rows_a, size=n), rng.integers(0, rows_b, size=n)
    main = sps.csr_matrix((np.ones(n), (np.arange(n), rng.integers(0, 20, size=n))), shape=(n, 20))
    blocks = [(ma, _multihot(rng, rows_a, cols_a, per_a)), (mb, _multihot(rng, rows_b, cols_b, per_b))]
    gi = np.concatenate([[0] * 20] + [[k + 1] * b.shape[1] for k, (_, b) in enumerate(blocks)]).astype(np.int32)
    y = rng.normal(size=n) + 0.3 * (ma % 7) - 0.2 * (mb % 5)
    return main, y, blocks, gi


def _chain(oracle, capi, design, rank=3, iters=2):
    main, y, blocks, gi = design
    t = oracle.OracleTrainer(main, y, blocks, rank=rank, group_index=gi)
    c = capi.Context(main, y, blocks, rank=rank, group_index=gi)
    c.set_state(*t.fm())
    c.set_e(t.e(main.shape[0]))
    drv = CapiGibbs(c, t.clone(), main.shape[0], gi)
    for it in range(iters):
        t.step()
        drv.step()
    return t, c


@pytest.mark.parametrize("cg,lw,nb,cap", [(4, 3, 16, 0), (4, 4, 16, 0), (2, 5, 8, 0), (1, 6, 4, 0), (4, 2, 32, 0), (3, 2, 5, 0), (4, 3, 16, 700), (2, 5, 16, 0), (4, 5, 8, 0), (2, 3, 8, 0), (2, 6, 32, 0), (1, 7, 16, 0)])
def test_streamed_chain_matches_oracle(oracle, capi, monkeypatch, cg, lw, nb, cap):
    monkeypatch.setenv("MFM_CHAIN_GRID_MIN", "0")  # (the test's blocks are small: their cold parts would not go to the grid otherwise)
    monkeypatch.setenv("MFM_NO_CELL", "1")         # the generic relation-block path (the cell path has its own tests)
    monkeypatch.setenv("MFM_CS_CG", str(cg))
    monkeypatch.setenv("MFM_CS_LW", str(lw))
    monkeypatch.setenv("MFM_CS_LW_MIN", "1")
    monkeypatch.setenv("MFM_CS_NB", str(nb))
    if cap:
        monkeypatch.setenv("MFM_CS_CAP", str(cap))
    design = _design()
    t, c = _chain(oracle, capi, design)
    assert c.plan_flags()["streamed_chain"]
    w0, w, V = t.fm()
    _, gw, gV = c.get_state()
    np.testing.assert_allclose(gV, V, rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(gw, w, rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(c.get_e(), t.e(design[0].shape[0]), rtol=1e-7, atol=1e-7)


def test_streamed_chain_is_reproducible_and_equals_the_batched_form(oracle, capi, monkeypatch):
    monkeypatch.setenv("MFM_CHAIN_GRID_MIN", "0")
    monkeypatch.setenv("MFM_NO_CELL", "1")
    design = _design(seed=8)
    _, c1 = _chain(oracle, capi, design)
    _, c2 = _chain(oracle, capi, design)
    assert c1.plan_flags()["streamed_chain"]
    assert np.array_equal(c1.get_state()[2], c2.get_state()[2]) and np.array_equal(c1.get_e(), c2.get_e())
    monkeypatch.setenv("MFM_NO_CB_STREAM", "1")
    _, c3 = _chain(oracle, capi, design)
    assert not c3.plan_flags()["streamed_chain"]
    np.testing.assert_allclose(c3.get_state()[2], c1.get_state()[2], rtol=1e-8, atol=1e-9)


def test_streamed_chain_on_the_cell_path_and_with_identity_columns(oracle, capi, monkeypatch):
    # an index-tuple design (the shape of BASELINE configs[4]): the blocks' feature sweeps run between the cell passes on the same records;
    # the user block carries an identity part (one-hot columns: a single entry each -- all cold or all hot) before its multi-hot part
    monkeypatch.setenv("MFM_CHAIN_GRID_MIN", "0")
    monkeypatch.setenv("MFM_CELL_MIN_ROWS", "0")
    main, blocks, y, shapes = ds.tuple_design(n_rows=80000, n_users=6000, n_items=3000, ctx=(40,), user_cols=60, item_cols=40, seed=6)
    gi = ds.group_index_from_shapes(shapes)
    t, c = _chain(oracle, capi, (main, y, blocks, gi), rank=3, iters=2)
    flags = c.plan_flags()
    assert flags["streamed_chain"] and flags["cell"], flags
    w0, w, V = t.fm()
    _, gw, gV = c.get_state()
    np.testing.assert_allclose(gV, V, rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(gw, w, rtol=1e-7, atol=1e-8)
