import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tests import datasets as ds
def run(no_cell):
    if no_cell: os.environ["MFM_NO_CELL"] = "1"
    else: os.environ.pop("MFM_NO_CELL", None)
    from myfm_amd import _myfm
    main, blocks, y, shapes = ds.config5_like(0.04, ordered=True)
    gi = ds.group_index_from_shapes(shapes)
    rels = [_myfm.RelationBlock(np.asarray(m, dtype=np.int64), B) for m, B in blocks]
    b = _myfm.ConfigBuilder()
    b.set_alpha_0(1.0).set_beta_0(1.0).set_gamma_0(1.0).set_mu_0(0.0).set_reg_0(1.0)
    b.set_group_index([int(g) for g in gi]).set_n_iter(6).set_n_kept_samples(1)
    b.set_task_type(_myfm.TaskType.ORDERED); b.set_cutpoint_groups([(5, np.arange(main.shape[0]))])
    p, h = _myfm.create_train_fm(16, 0.1, main, rels, y, 42, b.build(), lambda *a: False)
    fm = p.samples[-1]
    return np.array(fm.V), np.array(fm.w), np.array(fm.cutpoints[0]), [hh.alpha for hh in h.hypers]
a = run(False); b = run(True)
print("max |dV|", np.abs(a[0]-b[0]).max(), "max |dw|", np.abs(a[1]-b[1]).max(), "cut", a[2], b[2])
