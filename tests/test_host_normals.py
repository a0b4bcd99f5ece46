"""The bulk weight-initialisation normals (csrc/mfm_hostnormals.hpp) are, bit for bit, what one persistent
std::normal_distribution<double> on the trainer's std::mt19937 returns (FM::initialize_weight, FM.hpp:34-45), and leave the
engine where the plain loop leaves it."""
import numpy as np
import pytest

from myfm_amd import _myfm


@pytest.mark.parametrize("seed,discard,count,threads", [
    (42, 0, 1, 1), (42, 0, 2, 1), (42, 0, 7, 1), (7, 5, 1000, 1), (7, 623, 1001, 2), (3, 624, 4096, 4),
    (11, 1000, 100003, 4), (5, 17, 600001, 8), (1234, 0, 262144 * 2 + 5, 3),
])
def test_bulk_normals_are_the_plain_loop(seed, discard, count, threads):
    fast, plain, n1, n2 = _myfm.host_normals_selftest(seed, discard, count, 0.1, threads)
    assert np.array_equal(fast, plain)
    assert n1 == n2  # the engines continue with the same output
    assert np.isfinite(fast).all() and abs(fast.std() - 0.1) < 0.1 * (4.0 / np.sqrt(max(count, 2))) + (count < 50) * 1.0
