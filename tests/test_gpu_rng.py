"""The device random stream (csrc/mfm_rng.hpp) against the host's std::mt19937 + libstdc++
distributions (drawn through the oracle library, which uses them directly).

Accept/reject decisions are exact, so the variates line up one-to-one; a value may differ from the
host's in the last ulp because log() on the device is not glibc's (tolerance 4 ulp)."""
import numpy as np
import pytest
import scipy.sparse as sps

from . import datasets as ds

pytestmark = pytest.mark.gpu


def _host_program(t, ops):
    hv, zw, zv = [], {}, {}
    for kind, dest, count, offset, shape in ops:
        if kind == 1:
            vals = np.array([t.rng_gamma(shape, 1.0)])
        else:
            vals = t.rng_sample_normals(count)
        if dest == 0:
            hv.append(vals)
        elif dest == 1:
            zw[offset] = vals
        else:
            zv[offset] = vals
    return np.concatenate(hv), zw, zv


@pytest.mark.parametrize("seed,n_users,K", [(42, 700, 5), (7, 700, 5), (3, 9000, 7), (11, 60000, 8)])
def test_device_stream_matches_libstdcxx(oracle, seed, n_users, K):
    # (n_users=700, K=5): every op takes the single-workgroup path; (9000, 7): z_V has 63 k variates and
    # goes through the whole-GPU eval / scan / scatter path, z_w (9 k) stays on the small path;
    # (60000, 8): ~2.5 M engine outputs per iteration -> the parallel generator with jump-ahead (8 workgroups)
    from myfm_amd import _capi

    X, y, shapes = ds.onehot_mf(3000, n_users, 90, seed=1)
    D = X.shape[1]
    t = oracle.OracleTrainer(X, y, rank=K, seed=seed)  # generator state after initialize_weight
    c = _capi.Context(X, y, rank=K)
    st, pos = t.rng_state()
    c.rng_seed_mt19937(st, pos)
    ops = [
        (1, 0, 1, 0, (1.0 + 3000) / 2),  # alpha
        (0, 0, 1, 1, 0.0),               # w0
        (1, 0, 1, 2, 351.0),             # lambda_w
        (1, 0, 1, 3, 0.75),              # a shape < 1 (pow branch of gamma_distribution)
        (1, 0, 1, 4, 1.0),
        (0, 0, 3, 5, 0.0),               # mu_w
        (0, 1, D, 0, 0.0),               # z_w
        (1, 0, 1, 8, 45.5),              # lambda_V
        (0, 0, 2, 9, 0.0),               # mu_V
        (0, 2, K * D, 0, 0.0),           # z_V
    ]
    c.rng_set_program(ops)
    c.rng_prefetch()
    c.rng_prefetch()
    c.rng_prefetch()  # three sets in flight before the first acquire
    with pytest.raises(RuntimeError):
        c.rng_prefetch()
    for it in range(3):
        hv = c.rng_acquire()
        if it == 0:
            with pytest.raises(RuntimeError):  # the acquired set and the two in flight occupy the three slots
                c.rng_prefetch()
        else:
            c.rng_prefetch()  # the Gibbs loop's pattern: acquire, then produce the next set while this one is used
        zw, zv = c.rng_get_z()
        want_hv, want_zw, want_zv = _host_program(t, ops)
        for got, want in ((hv, want_hv), (zw, want_zw[0]), (zv.ravel(), want_zv[0])):
            assert got.shape == want.shape
            ulp = np.abs(got - want) / np.maximum(np.spacing(np.abs(want)), 1e-300)
            assert ulp.max() <= 4, (it, ulp.max(), np.argmax(ulp))
            assert (ulp == 0).mean() > 0.9


def test_device_rng_chain_matches_oracle(oracle):
    """full iterations with z == NULL: device variates + host-scaled hyper draws == oracle chain"""
    from myfm_amd import _capi

    from .gibbs_driver import CapiGibbs

    X, y, shapes = ds.onehot_mf(20000, 300, 40, seed=1)
    gi = ds.group_index_from_shapes(shapes)
    K = 4
    t = oracle.OracleTrainer(X, y, rank=K, group_index=gi)
    c = _capi.Context(X, y, rank=K, group_index=gi)
    c.set_state(*t.fm())
    c.set_e(t.e(X.shape[0]))
    drv = CapiGibbs(c, None, X.shape[0], gi)
    drv.use_device_rng(*t.rng_state())
    for it in range(4):
        t.step()
        drv.step()
        w0, w, V = t.fm()
        gw0, gw, gV = c.get_state()
        tol = 1e-9 if it == 0 else 1e-7
        assert abs(drv.w0 - w0) < tol
        np.testing.assert_allclose(gw, w, rtol=tol, atol=tol)
        np.testing.assert_allclose(gV, V, rtol=tol, atol=tol)
        assert abs(drv.alpha - t.hyper()["alpha"]) < tol * drv.alpha
