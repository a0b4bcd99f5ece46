"""BASELINE configs 2 and 4 (small, latency-bound shapes): GPU it/s through the estimator boundary vs the CPU oracle."""
import os, sys, time
import numpy as np
import scipy.sparse as sps
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myfm_amd import _myfm
from oracle import oracle as O
from tests import datasets as ds


def session(X, y, rels, gi, rank, seed=42):
    b = _myfm.ConfigBuilder()
    b.set_alpha_0(1.0).set_beta_0(1.0).set_gamma_0(1.0).set_mu_0(0.0).set_reg_0(1.0)
    b.set_group_index([int(g) for g in gi]).set_n_iter(10).set_n_kept_samples(0).set_task_type(_myfm.TaskType.REGRESSION)
    return _myfm.GibbsSession(rank, 0.1, X, rels, y, seed, b.build())


def timeit(step, sync, n_warm, n):
    for _ in range(n_warm):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    sync()
    return n / (time.perf_counter() - t0)


def config2():
    X, y, shapes = ds.movielens_like(80000, 943, 1682, rank_true=8, seed=0, user_offset=30.0, item_offset=20.0)
    gi = ds.group_index_from_shapes(shapes)
    s = session(X, y, [], gi, 8)
    gpu = timeit(s.step, s.synchronize, 5, 100)
    t = O.OracleTrainer(X, y, rank=8, group_index=gi)
    cpu = timeit(t.step, lambda: None, 2, 20)
    print("config 2 (ML-100k-shaped, rank 8): GPU %.1f it/s, CPU oracle %.1f it/s, plan %s" % (gpu, cpu, s.plan_info()))


def config4():
    # ML-100k-extended-shaped relation blocks
    rng = np.random.default_rng(0)
    N, nu, ni = 80000, 943, 1682
    pu = 1.0 / (np.arange(1, nu + 1) + 30.0); pi = 1.0 / (np.arange(1, ni + 1) + 20.0)
    u = rng.choice(nu, size=N, p=pu / pu.sum()); it = rng.choice(ni, size=N, p=pi / pi.sum())
    date = rng.integers(0, 212, size=N)
    main = sps.csr_matrix((np.ones(N), (np.arange(N), date)), shape=(N, 212))

    def multihot(n_rows, n_cols, mean):
        rows, cols, vals = [], [], []
        for r in range(n_rows):
            k = max(1, rng.poisson(mean))
            c = rng.choice(n_cols, size=min(k, n_cols), replace=False)
            rows += [r] * len(c); cols += list(c); vals += [1.0 / np.sqrt(len(c))] * len(c)
        return sps.csr_matrix((vals, (rows, cols)), shape=(n_rows, n_cols))

    def onehot(idx, n):
        return sps.csr_matrix((np.ones(len(idx)), (np.arange(len(idx)), idx)), shape=(len(idx), n))

    ub = sps.hstack([onehot(np.arange(nu), 944), onehot(rng.integers(0, 10, nu), 10), onehot(rng.integers(0, 21, nu), 21),
                     onehot(rng.integers(0, 10, nu), 10), multihot(nu, 1683, 85)]).tocsr()
    ib = sps.hstack([onehot(np.arange(ni), 1683), onehot(rng.integers(0, 10, ni), 10), multihot(ni, 19, 2),
                     multihot(ni, 944, 48)]).tocsr()
    shapes = [212, 944, 10, 21, 10, 1683, 1683, 10, 19, 944]
    gi = ds.group_index_from_shapes(shapes)
    y = np.clip(np.round(3.5 + rng.normal(size=N)), 1, 5)
    rels = [_myfm.RelationBlock([int(v) for v in u], ub), _myfm.RelationBlock([int(v) for v in it], ib)]
    s = session(main, y, rels, gi, 16)
    gpu = timeit(s.step, s.synchronize, 3, 30)
    t = O.OracleTrainer(main, y, [(u, ub), (it, ib)], rank=16, group_index=gi)
    cpu = timeit(t.step, lambda: None, 1, 10)
    print("config 4 (ML-100k-extended blocks, rank 16): GPU %.2f it/s, CPU oracle %.2f it/s, plan %s" % (gpu, cpu, s.plan_info()))
    s.timing_enable(True); s.timing_reset()
    for _ in range(3):
        s.step()
    tm = s.timing()
    for k, v in sorted(tm.items(), key=lambda kv: -kv[1][0])[:8]:
        print("   %-20s %8.3f ms/iter %6d launches/iter" % (k, v[0] / 3, v[1] // 3))


if __name__ == "__main__":
    config2()
    config4()
