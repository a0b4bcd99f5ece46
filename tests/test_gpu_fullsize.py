"""BASELINE config 3 at FULL size (N = 10 M rows, nnz = 20 M, D = 80 555, rank 32) through the C ABI: the CPU oracle
needs ~6 s per iteration there, so parity is checked through size-independent properties of the sampler instead:

  * the residual the sweeps maintain incrementally (e += h * delta per touched entry, FMTrainer.hpp:374) equals the
    residual update_e recomputes from scratch (:494) -- any lost, doubled or misplaced update of any of the
    2 x 10 M x 33 entry visits of an iteration shows up here;
  * q_train after update_V is X V[:, K-1] (:320, :373);
  * the recomputed residual is the closed-form FM score minus y (FM.hpp:78-135) on a row sample;
  * the chain is bit-reproducible (no order-dependent sums anywhere on the path).
"""
import numpy as np
import pytest

from . import datasets as ds
from .gibbs_driver import CapiGibbs

pytestmark = pytest.mark.gpu

N, NU, NI, K = 10_000_000, 69878, 10677, 32


@pytest.fixture(scope="module")
def capi():
    from myfm_amd import _capi

    if _capi.lib().mfm_device_count() < 1:
        pytest.fail("no HIP device visible: the gpu tests need a real MI355X")
    return _capi


@pytest.fixture(scope="module")
def design():
    X, y, shapes = ds.movielens_like(N, NU, NI, rank_true=32, seed=1)
    return X, y, ds.group_index_from_shapes(shapes)


def _start(capi, oracle, X, y, gi):
    t = oracle.OracleTrainer(X, y, rank=K, group_index=gi)  # initial weights, residual and generator state only
    c = capi.Context(X, y, rank=K, group_index=gi)
    c.set_state(*t.fm())
    c.set_e(t.e(N))
    drv = CapiGibbs(c, None, N, gi)
    drv.use_device_rng(*t.rng_state())
    return c, drv


def test_full_size_invariants(capi, oracle, design):
    X, y, gi = design
    c, drv = _start(capi, oracle, X, y, gi)
    flags = c.plan_flags()
    assert flags["soa"] and flags["fused_next"]  # the path bench.py measures
    seen = {}
    for it in range(2):
        drv.step(before_update_e=lambda: seen.update(e=c.get_e(), q=c.get_q()))
        e_new = c.get_e()
        w0, w, V = c.get_state()
        scale = np.abs(e_new).max()
        # incremental residual == recomputed residual (rounding of ~66 updates per row only)
        assert np.abs(seen["e"] - e_new).max() < 1e-9 * max(scale, 1.0)
        # q-cache of the last factor
        np.testing.assert_allclose(seen["q"], X @ V[:, K - 1], rtol=1e-11, atol=1e-12)
        # closed-form score on a row sample
        rows = np.random.default_rng(it).choice(N, size=200_000, replace=False)
        rows.sort()
        want = ds.fm_score(X[rows], w0, w, V) - y[rows]
        np.testing.assert_allclose(e_new[rows], want, rtol=1e-10, atol=1e-10)
        assert np.isfinite(V).all() and drv.alpha > 0
    v_first = V
    # bit-reproducible
    c2, drv2 = _start(capi, oracle, X, y, gi)
    for it in range(2):
        drv2.step()
    assert np.array_equal(c2.get_state()[2], v_first)
