"""debug: one factor of the resident latent sweep against the oracle, users and items separately"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MFM_SCATTER_MIN_NNZ", "1000")
from tests import datasets as ds
from oracle import oracle as orc
orc.build()
from myfm_amd import _capi as capi

n, nu, ni = int(os.environ.get("N", 150001)), 200, 120
X, y, shapes = ds.onehot_mf(n, nu, ni, seed=21, sort_by_user=True)
gi = ds.group_index_from_shapes(shapes)
rank = 3
t = orc.OracleTrainer(X, y, (), rank=rank, group_index=gi)
c = capi.Context(X, y, (), rank=rank, group_index=gi)
c.set_state(*t.fm())
c.set_e(t.e(n))
print("flags", c.plan_flags())
G, D = t.G, t.D
rng = np.random.default_rng(5)
lam = rng.uniform(0.5, 2.0, size=(G, rank))
mu = rng.normal(size=(G, rank)) * 0.1
h = t.hyper()
t.set_hyper(0.7, h["mu_w"], h["lambda_w"], mu, lam)
for f in range(rank):
    z = t.clone().rng_sample_normals(D)
    t.update_V_factor(f)
    c.sweep_V(f, f + 1, 0.7, lam, mu, z)
    V = t.fm()[2]
    gV = c.get_state()[2]
    du = np.abs(gV[:nu, f] - V[:nu, f])
    di = np.abs(gV[nu:, f] - V[nu:, f])
    de = np.abs(c.get_e() - t.e(n))
    print("factor", f, "users max err", du.max(), "n bad", (du > 1e-9).sum(), "items max err", di.max(), "n bad", (di > 1e-9).sum(),
          "e max err", de.max(), "n bad", (de > 1e-8).sum())
    if (du > 1e-9).any():
        print("  bad users", np.nonzero(du > 1e-9)[0][:20])
    if (di > 1e-9).any():
        print("  bad items", np.nonzero(di > 1e-9)[0][:20])
