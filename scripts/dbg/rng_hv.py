"""debug: the device stream's hyper variates against the oracle's generator, set by set"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import datasets as ds
from tests.test_gpu_rng import _host_program
from oracle import oracle as orc
orc.build()
from myfm_amd import _capi
seed, n_users, K = 3, 9000, 7
X, y, shapes = ds.onehot_mf(3000, n_users, 90, seed=1)
D = X.shape[1]
t = orc.OracleTrainer(X, y, rank=K, seed=seed)
c = _capi.Context(X, y, rank=K)
st, pos = t.rng_state()
c.rng_seed_mt19937(st, pos)
ops = [(1, 0, 1, 0, (1.0 + 3000) / 2), (0, 0, 1, 1, 0.0), (1, 0, 1, 2, 351.0), (1, 0, 1, 3, 0.75), (1, 0, 1, 4, 1.0), (0, 0, 3, 5, 0.0),
       (0, 1, D, 0, 0.0), (1, 0, 1, 8, 45.5), (0, 0, 2, 9, 0.0), (0, 2, K * D, 0, 0.0)]
c.rng_set_program(ops)
c.rng_prefetch(); c.rng_prefetch(); c.rng_prefetch()
for it in range(3):
    hv = c.rng_acquire()
    want_hv, _, _ = _host_program(t, ops)
    print(it, "got ", hv[:6]); print(it, "want", want_hv[:6])
