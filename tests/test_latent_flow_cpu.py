"""The algorithm behind the exact latent draws (csrc/mfm_latent.hip) on the CPU, in numpy: a sequential rejection chain
-- row t takes the first quad after row t - 1's that it accepts -- is one monotone lattice path over (row, quad); chunk maps computed
from ALL candidate entering rows of an a-priori window (walkers that meet merge) compose to the true path. Checked against the plain
sequential walk; no GPU, no product code (the prototype the kernels were written from: scripts/proto/flow_proto.py)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts", "proto"))


@pytest.mark.parametrize("n,lq", [(20000, 1024), (20000, 4096), (3000, 256)])
def test_chunk_maps_compose_to_the_sequential_path(n, lq):
    import flow_proto as fp

    rng = np.random.default_rng(3)
    p = rng.uniform(0.45, 0.95, n)
    took, jend = fp.sequential(p)
    R = fp.flows(p, lq, ksig=5.0, subq=max(64, lq // 4))
    assert R["ok"]
    for c, T in enumerate(R["entering"]):
        # the row in service at quad c * lq = rows served before it
        assert T == int(np.searchsorted(took, c * lq)), c
    # the walkers merge: far fewer candidate paths leave a chunk than enter it
    widest = max(f[1] - f[0] + 1 for f in R["finals"])
    assert R["maxlive_end"] * 4 < widest


def test_paths_are_monotone_and_coalesce():
    """two walkers of the same chunk never cross, and once they meet they stay together"""
    import flow_proto as fp

    rng = np.random.default_rng(5)
    p = rng.uniform(0.5, 0.9, 30000)
    a, b = np.int64(100), np.int64(102)  # (a gap of d rows closes after ~d^2 steps: the survivors of a window thin out as 1 / sqrt(steps))
    met = False
    for s in range(20000):
        j = np.array([1000 + s])
        a = a + fp.accept(np.array([a]), j, p)[0]
        b = b + fp.accept(np.array([b]), j, p)[0]
        assert a <= b
        if met:
            assert a == b
        met = met or a == b
    assert met
