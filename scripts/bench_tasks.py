"""Iteration rate of the three task types at the config-3 shape through the estimators (fit(3 n) - fit(n)) / 2 n."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from myfm_amd.utils import synthetic as ds
import myfm_amd

X, y, shapes = ds.movielens_like(10_000_000, 69878, 10677)
yc = (y > 3.5)
yo = np.clip(np.round(y), 1, 5).astype(np.int64) - 1
myfm_amd.MyFMRegressor(32).fit(X, y, group_shapes=shapes, n_iter=3, n_kept_samples=1)  # warm-up (module load, jump polynomials)
for name, est, yy in (("regression", myfm_amd.MyFMRegressor, y), ("classification", myfm_amd.MyFMClassifier, yc),
                      ("ordered probit", myfm_amd.MyFMOrderedProbit, yo)):
    ts = []
    for n in (10, 30):
        t = time.time()
        est(32).fit(X, yy, group_shapes=shapes, n_iter=n, n_kept_samples=1)
        ts.append(time.time() - t)
    print("%-16s %.1f it/s  (fit(10) %.2f s, fit(30) %.2f s)" % (name, 20 / (ts[1] - ts[0]), ts[0], ts[1]))
