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
    os.environ.setdefault("MFM_PLAN_CHECK", "1")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O

    O.build()
    return O
