"""BASELINE config 5 shape: N rows, main table = 2 one-hot fields, 4 relation blocks (user-side,
item-side, two small context blocks), ordered-probit target with 5 classes, MyFMOrderedProbit(rank).
`--scale 1.0` is the full shape (N = 50 M, nnz = 100 M, rank 64); default 0.1."""
import argparse, os, sys, time
import numpy as np
import scipy.sparse as sps
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=float, default=0.1)
ap.add_argument("--rank", type=int, default=64)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--task", default="ordered", choices=["ordered", "regression"])
a = ap.parse_args()

from myfm_amd import _myfm

rng = np.random.default_rng(2)
N = int(50_000_000 * a.scale)
nu, ni = max(1000, int(500_000 * a.scale)), max(200, int(50_000 * a.scale))
t0 = time.time()
u = np.sort(rng.integers(0, nu, size=N)).astype(np.int32)
it = rng.integers(0, ni, size=N).astype(np.int32)
indices = np.empty(2 * N, dtype=np.int32); indices[0::2] = u; indices[1::2] = nu + it
main = sps.csr_matrix((np.ones(2 * N), indices, np.arange(0, 2 * N + 1, 2, dtype=np.int64)), shape=(N, nu + ni))


def block(n_rows, n_cols, per_row):
    cols = rng.integers(0, n_cols, size=(n_rows, per_row))
    cols.sort(axis=1)
    # drop duplicate columns inside a row
    keep = np.ones_like(cols, dtype=bool); keep[:, 1:] = cols[:, 1:] != cols[:, :-1]
    rows = np.repeat(np.arange(n_rows), per_row).reshape(n_rows, per_row)
    return sps.csr_matrix((np.full(keep.sum(), 1.0 / np.sqrt(per_row)), (rows[keep], cols[keep])), shape=(n_rows, n_cols))


blocks = [(u.astype(np.int64), block(nu, 2000, 10)), (it.astype(np.int64), block(ni, 1000, 10)),
          (rng.integers(0, 1000, size=N), block(1000, 200, 5)), (rng.integers(0, 1000, size=N), block(1000, 200, 5))]
score = rng.normal(size=N) + 0.5 * np.sin(u * 0.01) + 0.3 * np.cos(it * 0.1)
y = np.digitize(score, np.quantile(score[: 1_000_000], [0.2, 0.4, 0.6, 0.8])).astype(np.float64)
shapes = [nu, ni] + [b.shape[1] for _, b in blocks]
gi = np.concatenate([np.full(s, g, dtype=np.int64) for g, s in enumerate(shapes)])
print("data %.1f s: N=%d nnz_main=%d D=%d" % (time.time() - t0, N, main.nnz, gi.size), flush=True)

b = _myfm.ConfigBuilder()
b.set_alpha_0(1.0).set_beta_0(1.0).set_gamma_0(1.0).set_mu_0(0.0).set_reg_0(1.0)
b.set_group_index([int(g) for g in gi]).set_n_iter(a.iters + 1).set_n_kept_samples(0)
if a.task == "ordered":
    b.set_task_type(_myfm.TaskType.ORDERED)
    b.set_cutpoint_groups([(5, list(range(N)))])
else:
    b.set_task_type(_myfm.TaskType.REGRESSION)
t0 = time.time()
rels = [_myfm.RelationBlock(m.tolist(), B) for m, B in blocks]
print("RelationBlock objects %.1f s" % (time.time() - t0), flush=True)
t0 = time.time()
s = _myfm.GibbsSession(a.rank, 0.1, main, rels, y, 42, b.build())
print("setup %.1f s, plan %s" % (time.time() - t0, s.plan_info()), flush=True)
s.step(); s.synchronize()
s.timing_enable(True); s.timing_reset()
t0 = time.perf_counter()
for _ in range(a.iters):
    s.step()
s.synchronize()
el = time.perf_counter() - t0
print("config-5 shape x%.2f, %s, rank %d: %.3f it/s (%.1f ms/iter)" % (a.scale, a.task, a.rank, a.iters / el, el / a.iters * 1e3))
tm = s.timing()
for k, v in sorted(tm.items(), key=lambda kv: -kv[1][0])[:10]:
    print("   %-20s %9.2f ms/iter %7d launches/iter %8.1f GB/s alg" % (k, v[0] / a.iters, v[1] // a.iters, v[2] / v[0] / 1e6))
