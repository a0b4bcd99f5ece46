"""myfm_amd -- MI355X-native Gibbs sampler for Bayesian Factorization Machines: a drop-in for the
MyFMRegressor / MyFMClassifier / MyFMOrderedProbit (.fit / .predict) + RelationBlock path of
tohtsky/myFM. The hot path runs in hand-written HIP kernels (libmyfm_hip.so, include/myfm_hip.h)
behind the pybind11 module ``myfm_amd._myfm``; there is no CPU fallback.

    from myfm_amd import MyFMRegressor, RelationBlock
"""
import os as _os

_here = _os.path.dirname(_os.path.abspath(__file__))
if not any(f.startswith("_myfm.") and f.endswith(".so") for f in _os.listdir(_here)):
    raise ImportError(
        "myfm_amd._myfm is not built. Run `python myfm_amd/_build.py` or `python -c 'import __graft_entry__ as g; g.build()'` "
        "(needs hipcc for gfx950 and g++). "
        "The package has no pure-Python / CPU fallback."
    )

from ._build import preload_hip_runtime as _preload  # noqa: E402

_hip_runtime = _preload()

from ._myfm import RelationBlock  # noqa: E402
from .estimators import (  # noqa: E402
    MyFMClassifier,
    MyFMGibbsClassifier,
    MyFMGibbsRegressor,
    MyFMOrderedProbit,
    MyFMRegressor,
)

__all__ = [
    "RelationBlock",
    "MyFMOrderedProbit",
    "MyFMRegressor",
    "MyFMClassifier",
    "MyFMGibbsRegressor",
    "MyFMGibbsClassifier",
]
