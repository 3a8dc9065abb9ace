"""Builds the in-tree gfx950 shared library (C ABI of include/flatnav_hip.h) with hipcc.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the dev container as well as on the MI355X box.
The kernels are templates over <element type, metric, row configuration>; their instantiations are compiled as 60
objects (kernel_inst.hip: 10 kernel families x 3 element types x 2 metrics) in parallel, plus beam_search.hip (host
code, C ABI, re-layout kernels), and linked into libflatnav_hip.so.  Objects are cached in csrc/_obj and rebuilt
when a source they include is newer.  The .so stays in-tree (git-ignored, but shipped by gpurun)."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libflatnav_hip.so")
HEADERS = [os.path.join(CSRC, f) for f in ("search_params.h", "kernel_table.h", "heaps.hpp", "distance.hpp", "visited.hpp",
                                           "kernels.hpp", "wire.hpp", "merged_beam.hpp", "relayout.hpp")] + [
    os.path.join(ROOT, "include", "flatnav", "util", "StlExact.h"), os.path.join(ROOT, "include", "flatnav_hip.h")]
MAIN = os.path.join(CSRC, "beam_search.hip")
INST = os.path.join(CSRC, "kernel_inst.hip")
TYPES = [("float", "f32"), ("uint8_t", "u8"), ("int8_t", "i8")]
METRICS = [(0, "l2"), (1, "ip")]
FAMILIES = [(0, "exact"), (3, "wire"), (4, "merged"), (5, "merged1"), (6, "merged0"), (7, "merged2"),
            (8, "merged_d"), (9, "merged1_d"), (10, "merged0_d"), (11, "merged2_d")]  # 8-11: the DIRECT forms (small launches)


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _units(defines):
    """(object path, source, extra -D flags) for every translation unit."""
    units = [(os.path.join(OBJ, "beam_search.o"), MAIN, [])]
    for ctype, tag in TYPES:
        for metric, mtag in METRICS:
            for fam, fname in FAMILIES:
                units.append((os.path.join(OBJ, "inst_%s_%s_%s.o" % (fname, tag, mtag)), INST,
                              ["-DFNV_INST_T=" + ctype, "-DFNV_INST_TAG=" + tag, "-DFNV_INST_METRIC=%d" % metric,
                               "-DFNV_INST_MTAG=" + mtag, "-DFNV_INST_FAMILY=%d" % fam]))
    return [(o, s, f + ["-D" + d for d in defines]) for o, s, f in units]


def _stale(obj, src):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    # relayout.hpp (upload / measurement kernels) is included by beam_search.hip only
    deps = HEADERS if src == MAIN else [h for h in HEADERS if not h.endswith("relayout.hpp")]
    return any(os.path.getmtime(d) > t for d in [src] + deps)


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in [MAIN, INST] + HEADERS)


def build(force: bool = False, verbose: bool = False, defines=(), out: str | None = None, jobs: int | None = None) -> str:
    out = out or LIB
    custom = bool(defines) or out != LIB  # profiling / experiment builds get their own object directory
    if not force and not custom and not needs_build():
        return LIB
    if custom:  # its own object directory: readable prefix + a hash of the whole define set and the output (no collisions)
        import hashlib

        tag = "_".join(sorted(d.replace("=", "-") for d in defines))
        objdir = OBJ + "_" + tag[:60] + "_" + hashlib.sha1((tag + "|" + out).encode()).hexdigest()[:10]
    else:
        objdir = OBJ
    os.makedirs(objdir, exist_ok=True)
    base = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-inline-asm",
            "-fvisibility=hidden", "-I" + os.path.join(ROOT, "include")]
    if verbose:
        base.insert(1, "-Rpass-analysis=kernel-resource-usage")
    units = [(os.path.join(objdir, os.path.basename(o)), s, f) for o, s, f in _units(defines)]
    if "FNV_DEV_FAST_BUILD" in defines:  # developer build: one translation unit, one instantiation per kernel
        units = units[:1]
    todo = [(o, s, f) for o, s, f in units if force or custom or _stale(o, s)]

    def compile_one(unit):
        o, s, f = unit
        cmd = base + f + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)

    jobs = jobs or max(1, min(8, os.cpu_count() or 1))
    with ThreadPoolExecutor(jobs) as pool:
        list(pool.map(compile_one, todo))
    subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-fvisibility=hidden"] +
                          [o for o, _, _ in units] + ["-o", out])
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose="-v" in sys.argv)
    print(LIB)
