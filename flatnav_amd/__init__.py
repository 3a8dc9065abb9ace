"""flatnav_amd -- MI355X-native batched k-NN search for flat navigable-small-world indexes.

Drop-in for the search path of BlaiseMuhirwa/flatnav with the reference's Python surface
(python-bindings/src/flatnav/__init__.py:1-35 of the reference):

    import flatnav_amd as flatnav
    index = flatnav.index.create(distance_type="l2", index_data_type=flatnav.data_type.DataType.float32,
                                 dim=128, dataset_size=N, max_edges_per_node=32)
    index.add(data=X, ef_construction=100)
    distances, labels = index.search(queries=Q, K=10, ef_search=100)     # runs on the GPU

Layers: `_core` (pybind11 over the header-only host API in include/flatnav/) -> C ABI
(include/flatnav_hip.h, libflatnav_hip.so) -> hand-written gfx950 HIP kernels.  `flatnav_amd.hip`
binds the C ABI directly with ctypes.  The search path has no CPU fallback.
"""
import sys as _sys

__version__ = "0.1.0"

from ._runtime import preload_hip_runtime as _preload_hip_runtime

_preload_hip_runtime()  # share torch's bundled HIP runtime when torch is installed (see _runtime.py)

try:
    from . import _core
except ImportError as _e:  # built in-tree by `python -m flatnav_amd.build_host`
    _core = None
    _core_error = _e

if _core is not None:
    from ._core import MetricType, data_type  # noqa: F401

    class _DataTypeModule:
        from ._core.data_type import DataType

    class _IndexModule:
        from ._core.index import (IndexIPFloat, IndexIPInt8, IndexIPUint8, IndexL2Float, IndexL2Int8, IndexL2Uint8,
                                  create)

    index = _IndexModule
    _sys.modules[__name__ + ".index"] = _IndexModule
    _sys.modules[__name__ + ".data_type"] = _DataTypeModule
    __all__ = ["MetricType", "data_type", "index", "__version__"]
else:

    def __getattr__(name):
        if name in ("index", "data_type", "MetricType"):
            raise ImportError("flatnav_amd._core is not built (run `python -m flatnav_amd.build_host`): %s"
                              % (_core_error,))
        raise AttributeError(name)
