import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from myfm_amd.utils import synthetic as ds
import myfm_amd
X, y, shapes = ds.movielens_like(10_000_000, 69878, 10677)
task = sys.argv[1] if len(sys.argv) > 1 else "classification"
if task == "classification":
    myfm_amd.MyFMClassifier(32).fit(X, y > 3.5, group_shapes=shapes, n_iter=20, n_kept_samples=1)
else:
    myfm_amd.MyFMOrderedProbit(32).fit(X, np.clip(np.round(y), 1, 5).astype(np.int64) - 1, group_shapes=shapes, n_iter=20, n_kept_samples=1)
