"""The configuration users and bench.py actually run, inside the driver's GPU suite.

tests/conftest.py puts every GPU test in CHECKER mode: MFM_PLAN_CHECK=1 (the host planners run beside the device ones and every
array is compared) and MFM_RES_MIN_ROWS=0 (small two-field tables take the persistent sweep too). Production mode -- device
planners only, generic plans built on demand (`plan_flags` 262 in the bench line), the persistent sweep only from 2^20 rows on --
is what this test re-runs a slice of the suite in: the golden-vector chains, the BASELINE workloads, the cell path and the
full-size chains, in a subprocess with MYFM_TEST_PRODUCTION=1 (see conftest.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SLICE = [
    "tests/test_golden_gpu.py",
    "tests/test_gpu_baseline_configs.py",
    "tests/test_gpu_cell.py",
    "tests/test_gpu_fullsize.py::test_full_size_invariants",
    "tests/test_gpu_fullsize.py::test_config3_full_size_one_rank_vs_oracle",
    "tests/test_gpu_fullsize.py::test_config5_full_size_n50m_rank64",
]


def test_production_mode_slice():
    if os.environ.get("MYFM_TEST_PRODUCTION"):
        pytest.skip("already the production-mode subprocess")
    env = dict(os.environ)
    for k in ("MFM_PLAN_CHECK", "MFM_RES_MIN_ROWS"):
        env.pop(k, None)
    env["MYFM_TEST_PRODUCTION"] = "1"
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + SLICE, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1500)
    tail = (r.stdout or "")[-3000:] + (r.stderr or "")[-1500:]
    assert r.returncode == 0, tail
    last = [l for l in r.stdout.splitlines() if " passed" in l]
    assert last and "failed" not in last[-1], tail
    print(last[-1])
