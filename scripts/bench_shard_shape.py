"""Single-GPU timing of the table ONE rank sees in an N-GPU weak-scaling run of bench.py (same users, N x rows:
users are N x longer, a growing share of them longer than a row tile)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myfm_amd import _myfm
from myfm_amd.utils import synthetic as ds

for world in (2, 4, 8):
    X, y, shapes, lo, total = ds.movielens_like_shard(10_000_000, world // 2, world, 69878, 10677)
    gi = ds.group_index_from_shapes(shapes)
    b = _myfm.ConfigBuilder()
    b.set_alpha_0(1.0).set_beta_0(1.0).set_gamma_0(1.0).set_mu_0(0.0).set_reg_0(1.0)
    b.set_group_index([int(g) for g in gi]).set_n_iter(40).set_n_kept_samples(0).set_task_type(_myfm.TaskType.REGRESSION)
    s = _myfm.GibbsSession(32, 0.1, X, [], y, 42, b.build())
    for _ in range(3):
        s.step()
    s.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        s.step()
    s.synchronize()
    el = time.perf_counter() - t0
    cnt = np.bincount(X.indices[0::2], minlength=69878)
    print("shard of world %d: %d rows, %d users present, %.1f%% of rows in users > 4096 rows: %.1f it/s (%.2f ms)"
          % (world, X.shape[0], (cnt > 0).sum(), 100.0 * cnt[cnt > 4096].sum() / X.shape[0], 20 / el, el / 20 * 1e3), flush=True)
