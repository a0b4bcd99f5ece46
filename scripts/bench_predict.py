"""Predictor::predict (predictor.hpp:35-147) at the config-3 shape: 95 kept samples (n_iter = 100), a 1 M-row test table.
Compares the device-resident sample store (samples read in place) with host-held samples (MYFM_AMD_HOST_SAMPLES=1: the
round-1 path, every sample uploaded for its scoring pass)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import myfm_amd  # noqa: E402
from myfm_amd.utils import synthetic as ds  # noqa: E402

X, y, shapes = ds.movielens_like(10_000_000, 69878, 10677, rank_true=32, seed=1)
Xt = X[np.sort(np.random.default_rng(0).choice(X.shape[0], size=1_000_000, replace=False))]
out = {}
for mode in ("device_store", "host_samples"):
    if mode == "host_samples":
        os.environ["MYFM_AMD_HOST_SAMPLES"] = "1"
    fm = myfm_amd.MyFMRegressor(32).fit(X, y, group_shapes=shapes, n_iter=100)
    fm.predict(Xt[:1000])  # warm-up
    t0 = time.perf_counter()
    p = fm.predict(Xt)
    out[mode] = (time.perf_counter() - t0, p)
    print("%-13s predict(1 M rows, %d samples): %.3f s" % (mode, len(fm.predictor_.samples), out[mode][0]), flush=True)
d = np.abs(out["device_store"][1] - out["host_samples"][1]).max()
print("max |difference| = %.3e   speed-up %.1fx" % (d, out["host_samples"][0] / out["device_store"][0]))
