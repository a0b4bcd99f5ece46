"""Kernel-time probe on an ML-10M-shaped design through the C ABI (numpy z, no parity): prints the
per-kernel-class HIP-event times, algorithmic GB/s and the share of one iteration."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myfm_amd import _capi
from myfm_amd.utils import synthetic as ds

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=10_000_000)
ap.add_argument("--users", type=int, default=69878)
ap.add_argument("--items", type=int, default=10677)
ap.add_argument("--rank", type=int, default=32)
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--unsorted", action="store_true")
ap.add_argument("--only", default="both", choices=["both", "users", "items"])
ap.add_argument("--ublock", type=int, default=0, help="re-sort rows by (user // ublock, item)")
ap.add_argument("--ml", action="store_true", help="ML-10M-like popularity (bench.py data)")
a = ap.parse_args()

t0 = time.time()
if a.ml:
    X, y, shapes = ds.movielens_like(a.rows, a.users, a.items, seed=1)
else:
    X, y, shapes = ds.onehot_mf(a.rows, a.users, a.items, rank_true=8, seed=1, sort_by_user=not a.unsorted)
if a.ublock:
    Xc = X.tocsr()
    u = Xc.indices[0::2].astype(np.int64)
    it = Xc.indices[1::2].astype(np.int64)
    order = np.lexsort((it, u // a.ublock))
    X = Xc[order]
    y = y[order]
if a.only == "users":
    X, shapes = X[:, : a.users].tocsr(), [a.users]
elif a.only == "items":
    X, shapes = X[:, a.users:].tocsr(), [a.items]
gi = ds.group_index_from_shapes(shapes)
print("data %.1fs  N=%d nnz=%d D=%d" % (time.time() - t0, X.shape[0], X.nnz, X.shape[1]), flush=True)
t0 = time.time()
c = _capi.Context(X, y, rank=a.rank, group_index=gi)
print("setup %.1fs plan=%s" % (time.time() - t0, c.plan_info()), flush=True)
rng = np.random.default_rng(0)
D, K, G = c.D, a.rank, len(shapes)
c.set_state(0.1, rng.normal(size=D) * 0.1, rng.normal(size=(D, K)) * 0.1)
c.update_e_regression()
lam_w, mu_w = np.ones(G), np.zeros(G)
lam_V, mu_V = np.ones((G, K)), np.zeros((G, K))
for it in range(a.iters + 1):
    if it == 1:
        c.timing_enable(True)
        c.timing_reset()
        c.synchronize()
        t0 = time.time()
    zw = rng.normal(size=D)
    zV = rng.normal(size=(K, D))
    se, se2 = c.reduce_e()
    c.shift_e(0.001)
    c.group_stats_w(mu_w)
    c.sweep_w(1.0, lam_w, mu_w, zw)
    c.group_stats_V(mu_V)
    c.sweep_V(0, K, 1.0, lam_V, mu_V, zV)
    c.update_e_regression()
c.synchronize()
wall = (time.time() - t0) / a.iters
tm = c.timing()
tot = sum(v[0] for v in tm.values()) / a.iters
print("wall/iter %.1f ms (incl. numpy RNG), kernel time/iter %.2f ms" % (wall * 1e3, tot))
for k, (ms, n, by) in sorted(tm.items(), key=lambda kv: -kv[1][0]):
    print("%-22s %9.3f ms/iter  %6d launches/iter  avg %8.1f us  %8.1f GB/s alg" % (k, ms / a.iters, n // a.iters, ms / n * 1e3, by / ms / 1e6))
