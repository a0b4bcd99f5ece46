"""The HIP path against the committed golden vectors (tests/golden/): the product binding (`_myfm`, device
random stream, z == NULL) must reproduce the seeded chains, and the device stream the libstdc++ variates."""
import numpy as np
import pytest

from .test_golden_cpu import CHAINS, fixture_design, load

pytestmark = pytest.mark.gpu


def test_device_stream_matches_libstdcxx_vectors():
    from myfm_amd import _capi

    from . import datasets as ds

    g = load("libstdcxx_stream")
    X, y = ds.toy()
    c = _capi.Context(X, y, rank=2)
    c.rng_seed_mt19937(g["state"], int(g["position"]))
    n = g["normals"].shape[0]
    ops = [(0, 0, n, 0, 0.0)] + [(1, 0, 1, n + i, float(s)) for i, s in enumerate(g["gamma_shapes"])] + \
          [(0, 0, 16, n + len(g["gamma_shapes"]), 0.0)]
    c.rng_set_program(ops)
    c.rng_prefetch()
    hv = c.rng_acquire()
    want = np.concatenate([g["normals"], g["gammas"], g["normals_after"]])
    ulp = np.abs(hv - want) / np.maximum(np.spacing(np.abs(want)), 1e-300)
    assert ulp.max() <= 4 and (ulp == 0).mean() > 0.9  # log() on the device is not glibc's: last-ulp differences


@pytest.mark.parametrize("name", CHAINS)
def test_product_chain_matches_vectors(name):
    from myfm_amd import _myfm

    g = load(name)
    X, y, blocks, gi, rank, seed = fixture_design(g)
    b = _myfm.ConfigBuilder()
    b.set_alpha_0(1.0).set_beta_0(1.0).set_gamma_0(1.0).set_mu_0(0.0).set_reg_0(1.0)
    b.set_group_index([int(v) for v in gi]).set_n_iter(int(g["iters"][-1])).set_n_kept_samples(0)
    b.set_task_type(_myfm.TaskType.REGRESSION)
    if name == "chain_blocks":
        b.set_fit_w0(False)
    rel = [_myfm.RelationBlock([int(v) for v in mp], B) for mp, B in blocks]
    sess = _myfm.GibbsSession(rank, 0.1, X, rel, y, seed, b.build())
    fm = sess.fm
    np.testing.assert_allclose(fm.w, g["init_w"], rtol=0, atol=0)
    np.testing.assert_allclose(fm.V, g["init_V"], rtol=0, atol=0)
    done = 0
    for it in g["iters"]:
        while done < it:
            sess.step()
            done += 1
        fm = sess.fm
        tol = 1e-7 if it <= 10 else 1e-5  # rounding differences of the summation order grow along the chain
        np.testing.assert_allclose(fm.w0, float(g["it%d_w0" % it]), rtol=tol, atol=tol * 1e-2)
        np.testing.assert_allclose(fm.w, g["it%d_w" % it], rtol=tol, atol=tol * 0.1)
        np.testing.assert_allclose(fm.V, g["it%d_V" % it], rtol=tol, atol=tol * 0.1)
        np.testing.assert_allclose(sess.hyper.alpha, float(g["it%d_alpha" % it]), rtol=tol * 0.1)
        np.testing.assert_allclose(sess.residual(), g["it%d_e" % it], rtol=tol, atol=tol)
