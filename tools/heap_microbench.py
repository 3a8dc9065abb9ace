#!/usr/bin/env python3
"""Developer tool: cycles per cooperative heap op (profiling build)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROF_LIB = os.path.join(ROOT, "flatnav_amd", "libflatnav_hip_prof.so")
L = C.CDLL(PROF_LIB)
L.fnv_debug_heap_microbench.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
out = (C.c_uint64 * 4)()
for size in (100, 300, 1000):
    for blocks in (1, 256, 1792):
        rc = L.fnv_debug_heap_microbench(size, 1, blocks, out)
        print("heap size %4d blocks %4d: push %4d cyc, pop %4d cyc, dependent ds_read %3d cyc (rc %d)" % (
            size, blocks, out[0], out[1], out[2], rc))
