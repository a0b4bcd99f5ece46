"""No kernel may depend on what the LDS held before it started.

`MFM_DEBUG_POISON_LDS=-1` (mfm_common.hpp) fills every CU's LDS with NaN bit patterns before each bracketed launch. A kernel
that reads a word it has not written -- even to multiply it by zero -- then shows up as NaN deterministically, instead of
only when a workgroup of the random-number side stream happened to leave such bits there (how the conflict-batched chain's
"lanes past the end read slot 0 with x = 0" was found: a batch without hot rows stages no slot 0). The chains must be
bit-identical with and without the poison. The switch is read once per process, hence the subprocesses.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import json, sys
import numpy as np
sys.path.insert(0, %(root)r)
from tests import datasets as ds
from myfm_amd import _myfm
task = sys.argv[1]
main, blocks, y, shapes = ds.config5_like(0.004, ordered=True)
N = main.shape[0]
if task == "cls":
    y = np.where(y >= 2, 1.0, -1.0)
rels = [_myfm.RelationBlock([int(v) for v in m], B) for m, B in blocks]
b = _myfm.ConfigBuilder()
b.set_alpha_0(1.0).set_beta_0(1.0).set_gamma_0(1.0).set_mu_0(0.0).set_reg_0(1.0)
b.set_group_index([int(g) for g in ds.group_index_from_shapes(shapes)]).set_n_iter(4).set_n_kept_samples(2)
b.set_task_type({"ord": _myfm.TaskType.ORDERED, "cls": _myfm.TaskType.CLASSIFICATION, "reg": _myfm.TaskType.REGRESSION}[task])
if task == "ord":
    b.set_cutpoint_groups([(5, np.arange(N))])
trace = []
def cb(i, fm, hyper, hist):
    trace.append([float(fm.w0), float(np.asarray(fm.w).sum()), float(np.asarray(fm.V).sum()), float(hyper.alpha)])
    return False
_myfm.create_train_fm(8, 0.1, main, rels, y, 42, b.build(), cb)
print("TRACE " + json.dumps(trace))
"""


def _run(task, poison):
    env = dict(os.environ)
    env.pop("MFM_DEBUG_POISON_LDS", None)
    if poison:
        env["MFM_DEBUG_POISON_LDS"] = "-1"
    out = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}, task], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("TRACE ")][-1]
    return json.loads(line[len("TRACE "):])


@pytest.mark.parametrize("task", ["reg", "cls", "ord"])
def test_relation_block_chain_is_independent_of_stale_lds(task):
    clean = _run(task, poison=False)
    poisoned = _run(task, poison=True)
    assert all(all(v == v and abs(v) < 1e300 for v in row) for row in poisoned), poisoned
    assert poisoned == clean
