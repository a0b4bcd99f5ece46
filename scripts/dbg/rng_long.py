"""debug: many iterations of the device random stream against the oracle's generator (hyper variates of every set)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import datasets as ds
from tests.test_gpu_rng import _host_program
from oracle import oracle as orc
orc.build()
from myfm_amd import _capi
seed, n_users, K = 3, int(os.environ.get("NU", 5000)), 8
X, y, shapes = ds.onehot_mf(3000, n_users, 90, seed=1)
D = X.shape[1]
t = orc.OracleTrainer(X, y, rank=K, seed=seed)
c = _capi.Context(X, y, rank=K)
st, pos = t.rng_state()
c.rng_seed_mt19937(st, pos)
ops = [(1, 0, 1, 0, (1.0 + 3000) / 2), (0, 0, 1, 1, 0.0), (1, 0, 1, 2, 351.0), (1, 0, 1, 3, 0.75), (0, 0, 3, 4, 0.0),
       (0, 1, D, 0, 0.0), (1, 0, 1, 7, 45.5), (0, 0, 2, 8, 0.0), (0, 2, K * D, 0, 0.0)]
c.rng_set_program(ops)
c.rng_prefetch(); c.rng_prefetch(); c.rng_prefetch()
bad = 0
for it in range(int(os.environ.get("NIT", 14))):
    hv = c.rng_acquire()
    c.rng_prefetch() if it else None
    zw, zv = c.rng_get_z()
    want_hv, want_zw, want_zv = _host_program(t, ops)
    ok = np.allclose(hv, want_hv, rtol=1e-12) and np.allclose(zv.ravel(), want_zv[0], rtol=1e-12) and np.allclose(zw, want_zw[0], rtol=1e-12)
    bad += not ok
    if not ok: print("set", it, "MISMATCH", np.abs(hv - want_hv).max(), np.abs(zv.ravel() - want_zv[0]).max())
print("sets with a mismatch:", bad)
