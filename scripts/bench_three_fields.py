"""Three one-hot fields (user-sorted, item, context), N = 10 M, rank 32: the generic form of the fused pass."""
import os, sys, time
import numpy as np
import scipy.sparse as sps
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myfm_amd import _myfm
from myfm_amd.utils import synthetic as ds

N, nu, ni, nc = 10_000_000, 69878, 10677, 1000
X, y, shapes = ds.movielens_like(N, nu, ni, rank_true=32, seed=1)
rng = np.random.default_rng(5)
ctx = rng.integers(0, nc, size=N).astype(np.int32)
ind = np.empty(3 * N, dtype=np.int32)
ind[0::3] = X.indices[0::2]
ind[1::3] = X.indices[1::2]
ind[2::3] = nu + ni + ctx
X3 = sps.csr_matrix((np.ones(3 * N), ind, np.arange(0, 3 * N + 1, 3, dtype=np.int64)), shape=(N, nu + ni + nc))
gi = ds.group_index_from_shapes([nu, ni, nc])
b = _myfm.ConfigBuilder()
b.set_alpha_0(1.0).set_beta_0(1.0).set_gamma_0(1.0).set_mu_0(0.0).set_reg_0(1.0)
b.set_group_index([int(g) for g in gi]).set_n_iter(40).set_n_kept_samples(0).set_task_type(_myfm.TaskType.REGRESSION)
s = _myfm.GibbsSession(32, 0.1, X3, [], y, 42, b.build())
print("plan", s.plan_info(), "flags", s.plan_flags(), flush=True)
for _ in range(3):
    s.step()
s.synchronize(); s.timing_enable(True); s.timing_reset()
t0 = time.perf_counter()
for _ in range(10):
    s.step()
s.synchronize()
el = time.perf_counter() - t0
tm = s.timing()
print("3 fields: %.1f it/s (%.2f ms)" % (10 / el, el / 10 * 1e3), {k: round(v[0] / 10, 3) for k, v in sorted(tm.items(), key=lambda kv: -kv[1][0])[:8]})
