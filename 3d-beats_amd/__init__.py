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

__all__ = ["DecisionTree", "DecisionForest", "LayeredDecisionForest", "DecisionTreeEvaluator", "DecisionTreeTrainer",
           "GpuBuffer", "HandPipeline", "HostFramesEvaluator",
           "DeviceArray", "HipRuntime", "MAX_UINT16", "RdfError", "device_ptr", "get_runtime", "set_runtime",
           "to_device", "host_mapped_array", "library_path", "synth"]
