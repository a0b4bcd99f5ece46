"""timing of Predictor.predict over the device store: repeated calls, the design built once per call by the estimator"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import myfm_amd
from tests import datasets as ds
X, y, shapes = ds.movielens_like(10_000_000, 69878, 10677, rank_true=32, seed=1)
Xt = X[np.sort(np.random.default_rng(0).choice(X.shape[0], size=1_000_000, replace=False))]
fm = myfm_amd.MyFMRegressor(32).fit(X, y, group_shapes=shapes, n_iter=100)
fm.predict(Xt[:1000])
for rep in range(3):
    t0 = time.perf_counter(); p = fm.predict(Xt); print("predict %.4f s" % (time.perf_counter() - t0), flush=True)
