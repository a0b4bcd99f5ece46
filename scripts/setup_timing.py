"""MFM_SETUP_TIMING=1 python scripts/setup_timing.py  -- stages of fit() setup at the config-3 shape (stderr of the library)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from myfm_amd.utils import synthetic as ds
import myfm_amd
X, y, shapes = ds.movielens_like(10_000_000, 69878, 10677)
for rep in range(2):
    t = time.time()
    myfm_amd.MyFMRegressor(32).fit(X, y, group_shapes=shapes, n_iter=2, n_kept_samples=1)
    print("fit(n_iter=2) total %.3f s" % (time.time() - t), file=sys.stderr)
