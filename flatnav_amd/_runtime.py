"""One HIP runtime per process.

PyTorch's ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64.so.1, and libflatnav_hip.so
needs a library with the same SONAME.  If our library is loaded first the system copy (/opt/rocm) gets
mapped, and torch later brings up a second HSA runtime that finds no GPU.  Loading torch's copy first
(when torch is installed) makes everything in the process -- torch, RCCL, our kernels -- share one runtime,
whatever the import order.  No torch: the system runtime is used as usual.
"""
from __future__ import annotations

import ctypes
import importlib.util
import os

_done = False


def preload_hip_runtime() -> None:
    global _done
    if _done:
        return
    _done = True
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass
