"""Builds the in-tree gfx950 shared library (C ABI of include/flatnav_hip.h) with hipcc.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the dev container as well as on
the MI355X box.  The .so stays in-tree (git-ignored, but shipped by gpurun)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libflatnav_hip.so")
SOURCES = [os.path.join(CSRC, "beam_search.hip")]
DEPS = SOURCES + [os.path.join(CSRC, f) for f in ("search_params.h", "heaps.hpp", "distance.hpp", "visited.hpp", "kernels.hpp", "wire.hpp", "fast_search.hpp")] + [
    os.path.join(ROOT, "include", "flatnav", "util", "StlExact.h"), os.path.join(ROOT, "include", "flatnav_hip.h")]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force: bool = False, verbose: bool = False, defines=(), out: str | None = None) -> str:
    out = out or LIB
    if not force and out == LIB and not needs_build():
        return LIB
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
           "-Wno-unused-function", "-fvisibility=hidden", "-I" + os.path.join(ROOT, "include")] + ["-D" + d for d in defines] + SOURCES + [
               "-o", out]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose="-v" in sys.argv)
    print(LIB)
