import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # every row-tile layout the planner packs on the device is also packed on the host and compared array by array
    # (mfm_plan.hpp build_scattered): a mismatch fails mfm_finalize of the test that built it
    # (MYFM_TEST_NO_PLAN_CHECK=1: the suite in production mode -- device planners only, generic plans built on demand; the planner
    #  fuzz tests, which ARE the comparison, skip themselves)
    # MYFM_TEST_PRODUCTION=1 (tests/test_gpu_production_mode.py re-runs a slice of the suite this way): neither of the two -- the
    # environment of a user's process and of bench.py
    production = bool(os.environ.get("MYFM_TEST_PRODUCTION"))
    if os.environ.get("MYFM_TEST_NO_PLAN_CHECK") or production:
        os.environ.pop("MFM_PLAN_CHECK", None)
    else:
        os.environ.setdefault("MFM_PLAN_CHECK", "1")
    # the persistent latent sweep (mfm_res.hpp) is the default only from 2^20 rows on; the tests' small two-field tables take it
    # too, so that every chain test of such a table also covers it (tests that want the per-factor passes set MFM_NO_RESIDENT)
    if not production:
        os.environ.setdefault("MFM_RES_MIN_ROWS", "0")
    # a fresh checkout has no built extension yet (the .so files are git-ignored): build in-tree once
    pkg = os.path.join(ROOT, "myfm_amd")
    if not any(f.startswith("_myfm.") and f.endswith(".so") for f in os.listdir(pkg)):
        import __graft_entry__ as g

        g.build()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O

    O.build()
    return O
