"""3d-beats_amd: MI355X-native randomized-decision-forest inference (the RDF hot path of 3d-beats).

The directory name is not a Python identifier; import it with
    rdf = importlib.import_module("3d-beats_amd")
or through the alias module `beats3d_amd` at the repository root.

Submodules mirror the reference's module names so that `decision_tree`, `util` and
`engine.buffer` can stand in for /root/reference/src/{decision_tree,util,engine/buffer}.py on the
inference path.
"""
from . import synth  # noqa: F401
from ._lib import RdfError, library_path  # noqa: F401
from .decision_tree import (DecisionForest, DecisionTree, DecisionTreeEvaluator,  # noqa: F401
                            DecisionTreeTrainer, LayeredDecisionForest)
from .device import DeviceArray, HipRuntime, device_ptr, get_runtime, host_mapped_array, set_runtime, to_device  # noqa: F401
from .engine.buffer import GpuBuffer  # noqa: F401
from .host_stream import HostFramesEvaluator  # noqa: F401
from .pipeline import HandPipeline  # noqa: F401
from .util import MAX_UINT16  # noqa: F401

_REFERENCE_MODULE_NAMES = ("decision_tree", "util", "engine", "engine.buffer", "cuda", "cuda.points_ops", "cuda.mean_shift",
                           "cuda.py_nvcc_utils")


def install_reference_aliases(force=False):
    """Make the reference's own import lines resolve to this package: after this call `from decision_tree import *`,
    `from cuda.points_ops import *`, `import cuda.py_nvcc_utils as py_nvcc_utils`, `from cuda.mean_shift import *`,
    `from engine.buffer import GpuBuffer` and `from util import MAX_UINT16` (run_live_layered.py:6-14, 3d_bz.py:1-20) import
    the modules of `3d-beats_amd`.  (The package's modules import each other relatively, so putting its directory on
    `sys.path` is not enough: the names are registered in `sys.modules`.)  A name that is already imported from somewhere else
    is left alone and reported, unless `force`.  Returns the list of names installed."""
    import importlib
    import sys
    done = []
    for name in _REFERENCE_MODULE_NAMES:
        mod = importlib.import_module(f"{__name__}.{name}")
        have = sys.modules.get(name)
        if have is not None and have is not mod and not force:
            raise ImportError(f"install_reference_aliases: a module named {name!r} is already imported from "
                              f"{getattr(have, '__file__', '?')}; pass force=True to replace it")
        sys.modules[name] = mod
        done.append(name)
    return done


__all__ = ["DecisionTree", "DecisionForest", "LayeredDecisionForest", "DecisionTreeEvaluator", "DecisionTreeTrainer",
           "GpuBuffer", "HandPipeline", "HostFramesEvaluator",
           "DeviceArray", "HipRuntime", "MAX_UINT16", "RdfError", "device_ptr", "get_runtime", "set_runtime",
           "to_device", "host_mapped_array", "library_path", "synth", "install_reference_aliases"]
