"""mfm_regression_iteration (the default; MYFM_AMD_DEVICE_HYPERS=0 switches it off): a regression chain on the persistent sweep draws its hyper-parameters on
the device and enqueues a whole iteration at once. The chain must not depend on it: the default form (statistics read back, draws
on the host, mfm_sweep_wV) gives the same kept samples and the same hyper-parameter trajectory, bit for bit."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys
import numpy as np
import myfm_amd.utils.synthetic as ds
from myfm_amd import MyFMRegressor
X, y, gs = ds.movielens_like(60000, 700, 300, seed=5)
fm = MyFMRegressor(rank=6, random_seed=11, fit_w0=(sys.argv[2] == "1")).fit(X, y, group_shapes=gs, n_iter=9, n_kept_samples=6)
tr = fm.get_hyper_trace()
from myfm_amd import _myfm
out = {"hyper": np.asarray(tr.to_numpy(), dtype=float), "device_iterations": np.array([_myfm.device_hyper_iterations()])}
for i, s in enumerate(fm.predictor_.samples):
    out["w0_%d" % i] = np.array([s.w0]); out["w_%d" % i] = np.asarray(s.w); out["V_%d" % i] = np.asarray(s.V)
np.savez(sys.argv[1], **out)
"""


def _run(tmp_path, name, fit_w0, host):
    env = dict(os.environ)
    env["MFM_RES_MIN_ROWS"] = "0"  # (a small two-field table takes the persistent sweep)
    env.pop("MFM_PLAN_CHECK", None)  # (checker mode keeps update_e on the row-order scorer: no slot-order sums, no device iteration)
    env.pop("MYFM_AMD_DEVICE_HYPERS", None)  # (the device form is the default)
    if host:
        env["MYFM_AMD_DEVICE_HYPERS"] = "0"
    out = str(tmp_path / (name + ".npz"))
    r = subprocess.run([sys.executable, "-c", SCRIPT, out, "1" if fit_w0 else "0"], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, (r.stdout or "")[-2000:] + (r.stderr or "")[-3000:]
    return dict(np.load(out))


@pytest.mark.parametrize("fit_w0", [True, False])
def test_device_hyper_draws_equal_the_host_trainers(tmp_path, fit_w0):
    dev = _run(tmp_path, "dev", fit_w0, host=False)
    host = _run(tmp_path, "host", fit_w0, host=True)
    assert dev.keys() == host.keys() and len(dev) == 2 + 3 * 6
    assert int(dev.pop("device_iterations")[0]) >= 7 and int(host.pop("device_iterations")[0]) == 0  # (the first iteration starts from a row-order residual)
    assert np.isfinite(dev["hyper"]).all() and dev["hyper"].shape[0] == 9
    for k in dev:
        np.testing.assert_array_equal(dev[k], host[k], err_msg=k)
