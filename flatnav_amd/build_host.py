"""Builds `flatnav_amd/_core` -- the pybind11 module over the header-only host API (include/flatnav/)
-- with g++, linked against the in-tree libflatnav_hip.so (rpath $ORIGIN)."""
from __future__ import annotations

import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "python_module.cpp")
OUT = os.path.join(HERE, "_core" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def _deps():
    deps = [SRC, os.path.join(ROOT, "include", "flatnav_hip.h")]
    for d, _, files in os.walk(os.path.join(ROOT, "include", "flatnav")):
        deps += [os.path.join(d, f) for f in files]
    return deps


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in _deps())


def build(force: bool = False) -> str:
    from flatnav_amd import build as hip_build

    hip_build.build()
    if not force and not needs_build():
        return OUT
    import pybind11

    # No -ffast-math and no FP contraction: the host builder's distance order is the documented one.
    cmd = ["g++", "-O3", "-std=c++17", "-march=x86-64-v3", "-ffp-contract=off", "-fPIC", "-shared", "-pthread",
           "-fvisibility=hidden", "-I" + os.path.join(ROOT, "include"), "-I" + pybind11.get_include(),
           "-I" + sysconfig.get_paths()["include"], SRC, "-o", OUT, "-L" + HERE, "-lflatnav_hip",
           "-Wl,-rpath,$ORIGIN"]
    subprocess.check_call(cmd)
    return OUT


def build_program(src: str, out: str, force: bool = False) -> str:
    """Compile a C++ program against the header-only host API and the in-tree libflatnav_hip.so (rpath to it)."""
    from flatnav_amd import build as hip_build

    hip_build.build()
    deps = [d for d in _deps() if not d.endswith("python_module.cpp")] + [src]
    if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    cmd = ["g++", "-O2", "-std=c++17", "-march=x86-64-v3", "-ffp-contract=off", "-pthread", "-I" + os.path.join(ROOT, "include"),
           src, "-o", out, "-L" + HERE, "-lflatnav_hip", "-Wl,-rpath," + HERE]
    subprocess.check_call(cmd)
    return out


TOOLS = {"flatnav_construct": os.path.join(ROOT, "tools", "flatnav_construct.cpp"),
         "flatnav_query": os.path.join(ROOT, "tools", "flatnav_query.cpp")}


def build_tools(force: bool = False) -> dict:
    """The construct / query command-line pair (reference tools/construct_npy.cpp, tools/query_npy.cpp)."""
    return {name: build_program(src, os.path.join(ROOT, "tools", name + ".bin"), force) for name, src in TOOLS.items()}


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    if "--tools" in sys.argv:
        print(build_tools(force="--force" in sys.argv))
