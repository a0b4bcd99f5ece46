import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import datasets as ds
from myfm_amd import _myfm
main, blocks, y, shapes = ds.config5_like(0.004, ordered=True)
N = main.shape[0]
gi = ds.group_index_from_shapes(shapes)
rels = [] if os.environ.get("NOBLK") else [_myfm.RelationBlock([int(v) for v in m], B) for m, B in blocks]
TASK = os.environ.get("TASK", "ord")
if TASK == "cls": y = np.where(y >= 2, 1.0, -1.0)
b = _myfm.ConfigBuilder()
b.set_alpha_0(1.0).set_beta_0(1.0).set_gamma_0(1.0).set_mu_0(0.0).set_reg_0(1.0)
b.set_group_index([int(g) for g in gi]).set_n_iter(int(os.environ.get("NIT", 6))).set_n_kept_samples(2)
b.set_task_type({"ord": _myfm.TaskType.ORDERED, "cls": _myfm.TaskType.CLASSIFICATION, "reg": _myfm.TaskType.REGRESSION}[TASK])
if TASK == "ord": b.set_cutpoint_groups([(5, np.arange(N))])
def cb(i, fm, hyper, hist):
    print("iter", i, "alpha", hyper.alpha, "w0", fm.w0, "lam_w", np.asarray(hyper.lambda_w)[:3], "V finite", np.isfinite(np.asarray(fm.V)).all(), "cut", (fm.cutpoints[0] if len(fm.cutpoints) else None), flush=True)
    return False
try:
    p, h = _myfm.create_train_fm(8, 0.1, main, rels, y, 42, b.build(), cb)
    print("ok")
except Exception as e:
    print("EXC", e)
