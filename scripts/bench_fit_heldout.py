import time, numpy as np, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from myfm_amd.utils import synthetic as ds
import myfm_amd
X, y, shapes = ds.movielens_like(10_000_000, 69878, 10677)
n_test = 1_000_000
Xt, yt = X[:n_test], y[:n_test]
for n_iter in (20, 60):
    t = time.time()
    fm = myfm_amd.MyFMRegressor(32).fit(X, y, group_shapes=shapes, n_iter=n_iter, n_kept_samples=1, X_test=Xt, y_test=yt)
    print("n_iter", n_iter, "seconds", round(time.time() - t, 3))
