"""Seeded random one-hot designs through the C ABI against the oracle chain: number of fields, field sizes (with
unused features), row order, values, rank, tile size and the scatter threshold vary, so that every combination of the
plan's fast paths (binned / tile levels, split layout, fused pass and its two-field / multi-level / long-column
forms, columns without entries) gets exercised on shapes nobody hand-picked."""
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


def _random_design(seed):
    rng = np.random.default_rng(seed)
    n_fields = int(rng.integers(1, 5))
    n = int(rng.integers(3000, 70000))
    sizes = [int(rng.integers(3, 400)) for _ in range(n_fields)]
    used = [max(2, int(sz * rng.uniform(0.5, 1.0))) for sz in sizes]  # the rest of a field never occurs
    cols, off = [], 0
    for sz, us in zip(sizes, used):
        p = 1.0 / (np.arange(1, us + 1) + rng.uniform(0.5, 30.0))
        ids = rng.permutation(sz)[:us]
        cols.append(off + ids[rng.choice(us, size=n, p=p / p.sum())])
        off += sz
    C = np.stack(cols, axis=1)
    if rng.random() < 0.7:  # most tables arrive sorted by their first field
        C = C[np.argsort(C[:, 0], kind="stable")]
    data = np.ones(n * n_fields) if rng.random() < 0.6 else rng.choice([0.5, 1.0, 1.5, -1.0], size=n * n_fields)
    X = sps.csr_matrix((data, C.ravel().astype(np.int32), np.arange(0, n * n_fields + 1, n_fields)), shape=(n, off))
    X.sort_indices()
    y = rng.normal(size=n) + 0.3 * (C[:, 0] % 3)
    rank = int(rng.integers(1, 6))
    env = {"MFM_SCATTER_MIN_NNZ": str(int(rng.choice([500, 4000, 1 << 30]))), "MFM_TILE_BITS": str(int(rng.choice([9, 10, 12])))}
    return X, y, ds.group_index_from_shapes(sizes), rank, env


@pytest.mark.parametrize("seed", list(range(16)))
def test_random_design_matches_oracle(oracle, capi, monkeypatch, seed):
    X, y, gi, rank, env = _random_design(seed)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    t = oracle.OracleTrainer(X, y, rank=rank, group_index=gi)
    c = capi.Context(X, y, rank=rank, group_index=gi)
    c.set_state(*t.fm())
    c.set_e(t.e(X.shape[0]))
    drv = CapiGibbs(c, t.clone(), X.shape[0], gi)
    for it in range(3):
        t.step()
        drv.step()
        np.testing.assert_allclose(c.get_state()[2], t.fm()[2], rtol=1e-7, atol=1e-8, err_msg=str((seed, it, c.plan_flags(), env)))
        np.testing.assert_allclose(c.get_state()[1], t.fm()[1], rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(c.get_e(), t.e(X.shape[0]), rtol=1e-7, atol=1e-7)
