import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from myfm_amd import _myfm
from tests import datasets as ds
X, y, shapes = ds.movielens_like(10_000_000, 69878, 10677, rank_true=32, seed=1)
gi = ds.group_index_from_shapes(shapes)
for task, yy in (("CLASSIFICATION", (y > 3.5).astype(np.float64)), ("ORDERED", np.clip(np.floor(y) - 1, 0, 4))):
    b = _myfm.ConfigBuilder()
    b.set_alpha_0(1.0).set_beta_0(1.0).set_gamma_0(1.0).set_mu_0(0.0).set_reg_0(1.0)
    b.set_group_index([int(g) for g in gi]).set_n_iter(40).set_n_kept_samples(0).set_task_type(getattr(_myfm.TaskType, task))
    if task == "ORDERED":
        b.set_cutpoint_groups([(5, list(range(X.shape[0])))])
    s = _myfm.GibbsSession(32, 0.1, X, [], yy, 42, b.build())
    for _ in range(3): s.step()
    s.synchronize(); s.timing_enable(True); s.timing_reset()
    t0 = time.perf_counter()
    for _ in range(10): s.step()
    s.synchronize(); el = time.perf_counter() - t0
    tm = s.timing()
    print(task, "%.1f it/s (%.2f ms)" % (10 / el, el / 10 * 1e3), {k: round(v[0] / 10, 3) for k, v in sorted(tm.items(), key=lambda kv: -kv[1][0])[:6]}, flush=True)
