"""the layouts mfm_finalize builds ON THE DEVICE -- the persistent sweep's slot layout (mfm_res_plan.hpp), the cell
plan (mfm_cell.hip), level schedules / first-level scans / conflict batches (mfm_plan.hpp), the blocks' inverse maps
(mfm_block_kernels.hpp) -- against the host builders they replaced, array for array (MFM_PLAN_CHECK=1: mfm_finalize builds both
and raises on the first array that differs), over seeded random shapes incl. the awkward ones (a single user, users that never
occur, very long and very short item runs, workgroup counts 1 ... all CUs, 32-bit item indices, blocks that share streams)."""
import numpy as np
import pytest
import scipy.sparse as sps

from . import datasets as ds

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from myfm_amd import _capi

    if _capi.lib().mfm_device_count() < 1:
        pytest.fail("no HIP device visible: the gpu tests need a real MI355X")
    return _capi


TAKEN = []


def _two_field(rng, n_rows, n_users, n_items, zipf, gaps):
    """user-sorted two-field table; gaps: leave some user / item columns without rows"""
    pu = 1.0 / np.arange(1, n_users + 1) ** (0.5 * zipf)
    pi = 1.0 / np.arange(1, n_items + 1) ** zipf
    u = np.sort(rng.permutation(n_users)[rng.choice(n_users, size=n_rows, p=pu / pu.sum())])
    i = rng.permutation(n_items)[rng.choice(n_items, size=n_rows, p=pi / pi.sum())]
    pad_u = int(gaps * n_users)
    pad_i = int(gaps * n_items)
    indices = np.empty(2 * n_rows, dtype=np.int32)
    indices[0::2] = u
    indices[1::2] = n_users + pad_u + i
    X = sps.csr_matrix((np.ones(2 * n_rows), indices, np.arange(0, 2 * n_rows + 1, 2, dtype=np.int64)),
                       shape=(n_rows, n_users + pad_u + n_items + pad_i))
    return X, rng.normal(size=n_rows), [n_users + pad_u, n_items + pad_i]


@pytest.mark.parametrize("seed", range(12))
def test_resident_layout_device_equals_host(capi, monkeypatch, seed):
    if __import__("os").environ.get("MFM_PLAN_CHECK") is None:
        pytest.skip("needs the checker mode (MFM_PLAN_CHECK=1, the default of tests/conftest.py)")
    rng = np.random.default_rng(1000 + seed)
    cus = [None, 1, 3, 17, 64][seed % 5]
    n_rows = int(rng.integers(2000, min(200000, 30000 * (cus or 256))))
    n_users = int(rng.integers(1, 1 + [400, 3000][seed % 4 == 3]))  # (a workgroup holds at most 512 users: some shapes are refused)
    n_items = int(rng.integers(1, 1 + [400, 20000][seed % 3 == 2]))  # (every third shape: often more items than the workgroups can draw)
    if cus:
        monkeypatch.setenv("MFM_RES_CUS", str(cus))
    monkeypatch.setenv("MFM_SCATTER_MIN_NNZ", "1000")
    X, y, shapes = _two_field(rng, n_rows, n_users, n_items, float(rng.uniform(0.0, 1.5)), float(rng.choice([0.0, 0.3])))
    gi = ds.group_index_from_shapes(shapes)
    c = capi.Context(X, y, rank=2, group_index=gi)  # (raises "plan check: ..." when a device-built array differs from the host's)
    TAKEN.append(bool(c.plan_flags()["resident"]))  # (a refusal -- e.g. more items than the workgroups can draw -- must be the same on both sides)
    if seed == 11:
        assert sum(TAKEN) >= 6, TAKEN  # most of these shapes do take the persistent sweep


@pytest.mark.parametrize("seed", range(12))
def test_cell_plan_device_equals_host(capi, monkeypatch, seed):
    rng = np.random.default_rng(2000 + seed)
    monkeypatch.setenv("MFM_CELL_MIN_ROWS", "0")
    monkeypatch.setenv("MFM_CELL_GROUPS", str(int(rng.choice([1, 2, 7, 40, 256]))))
    n_rows = int(rng.integers(3000, 120000))
    kw = dict(n_rows=n_rows, n_users=int(rng.integers(1, 4000)), n_items=int(rng.integers(2, 90000)),
              ctx=tuple(int(x) for x in rng.integers(1, 200, size=int(rng.integers(0, 3)))), seed=int(seed),
              third_field=int(rng.choice([0, 0, 9])), with_item_field=bool(rng.integers(0, 2)), with_item_block=bool(rng.integers(0, 2)),
              with_user_block=bool(rng.integers(0, 2)))
    if not kw["with_item_field"] and not kw["with_item_block"] and not kw["with_user_block"] and not kw["ctx"]:
        kw["with_user_block"] = True
    main, blocks, y, shapes = ds.tuple_design(**kw)
    gi = ds.group_index_from_shapes(shapes)
    c = capi.Context(main, y, blocks, rank=2, group_index=gi)  # (both planners run; a difference raises)
    assert isinstance(c.plan_flags()["cell"], bool)


@pytest.mark.parametrize("seed", range(8))
def test_block_chains_levels_and_inverse_maps_device_equal_host(capi, monkeypatch, seed):
    # generic relation-block path (no cell path): multi-hot blocks with deep level schedules, conflict batches forced on small
    # blocks, scattered and sorted maps
    rng = np.random.default_rng(3000 + seed)
    monkeypatch.setenv("MFM_NO_CELL", "1")
    monkeypatch.setenv("MFM_CHAIN_FORCE_BATCHED", "1")
    monkeypatch.setenv("MFM_CHAIN_HOT_CAP", str(int(rng.choice([64, 200, 1200]))))
    if seed % 2:
        monkeypatch.setenv("MFM_CHAIN_GRID_MIN", "1")
    n_rows = int(rng.integers(5000, 60000))
    main, blocks, y, shapes = ds.tuple_design(n_rows=n_rows, n_users=int(rng.integers(50, 3000)), n_items=int(rng.integers(50, 5000)),
                                              ctx=(int(rng.integers(2, 90)),), user_cols=int(rng.integers(5, 300)),
                                              item_cols=int(rng.integers(5, 200)), seed=100 + seed)
    gi = ds.group_index_from_shapes(shapes)
    c = capi.Context(main, y, blocks, rank=2, group_index=gi)
    assert not c.plan_flags()["cell"]
