"""Time the device random-stream kernels alone (no concurrent sweeps)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myfm_amd import _capi
from myfm_amd.utils import synthetic as ds

X, y, shapes = ds.onehot_mf(100000, 69878, 10677, seed=1)
K, D = 32, X.shape[1]
c = _capi.Context(X, y, rank=K)
rs = np.random.RandomState(0)
c.rng_seed_mt19937(rs.randint(0, 2**32, size=624, dtype=np.uint64).astype(np.uint32), 624)
c.rng_set_program([(1, 0, 1, 0, 5.0), (0, 1, D, 0, 0.0), (0, 2, K * D, 0, 0.0)])
for it in range(4):
    t0 = time.perf_counter()
    c.rng_prefetch()
    c.rng_acquire()
    print("prefetch+acquire %.2f ms" % ((time.perf_counter() - t0) * 1e3))
