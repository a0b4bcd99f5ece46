"""The seeded input generators live in the package (`myfm_amd/utils/synthetic.py`: bench.py and smoke() use them too); the tests
import them from here."""
from myfm_amd.utils.synthetic import *  # noqa: F401,F403
from myfm_amd.utils.synthetic import _multihot, _onehot  # noqa: F401
