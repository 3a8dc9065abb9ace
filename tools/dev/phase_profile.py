#!/usr/bin/env python3
"""Developer tool: per-phase shader-cycle breakdown of beam_search_kernel on the bench workload.

Builds a profiling variant of the library (-DFNV_PHASE_TIMING -> flatnav_amd/libflatnav_hip_prof.so),
runs the bench-shaped search and prints cycles per phase per query/hop.  Not part of the product."""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
PROF_LIB = os.path.join(ROOT, "flatnav_amd", "libflatnav_hip_prof.so")
PHASES = ["setup", "entry_scan", "pop / select", "link_row", "visited", "distances", "admission(other)", "finalize",
          "candpop.choices", "candpop.chase", "candpop.fix", "candpush", "nbrpush", "nbrpop.choices", "nbrpop.chase",
          "nbrpop.fix"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--nq", type=int, default=10_000)
    ap.add_argument("--ef", type=int, default=100)
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--dtype", default="float32", choices=["float32", "uint8"], help="element type of the 128-d index")
    ap.add_argument("--dim768", action="store_true", help="768-d float32 inner product on low-rank unit vectors (config C3's shape)")
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--define", action="append", default=[], help="extra -D for the profiling build (e.g. FNV_NO_SPEC_ROW)")
    ap.add_argument("--tag", default="", help="suffix of the profiling library's file name (one per set of --define)")
    args = ap.parse_args()
    from flatnav_amd import build as hb

    prof_lib = PROF_LIB if args.dtype == "float32" else PROF_LIB.replace(".so", "_u8.so")
    if args.dim768:
        prof_lib = PROF_LIB.replace(".so", "_768.so")
    if args.tag:
        prof_lib = prof_lib.replace(".so", "_%s.so" % args.tag)
    if not os.path.exists(prof_lib) or args.build_only:
        # one instantiation only: float / L2 / 128-d (G=8, CU=4), or uint8 / L2 / 128-d (G=8, CU=1)
        extra = [] if args.dtype == "float32" else ["FNV_DEV_T=uint8_t", "FNV_DEV_CU=1"]
        if args.dim768:
            extra = ["FNV_DEV_G=64", "FNV_DEV_CU=3", "FNV_DEV_METRIC=FNV_METRIC_IP"]
        hb.build(force=True, defines=["FNV_PHASE_TIMING", "FNV_DEV_FAST_BUILD"] + extra + args.define, out=prof_lib)
    if args.build_only:
        return
    os.environ["FLATNAV_HIP_LIB"] = prof_lib
    import numpy as np

    import flatnav_amd as flatnav
    from flatnav_amd import datasets as ds
    from flatnav_amd import hip

    DIM, metric = (768, "angular") if args.dim768 else (128, "l2")
    if args.dim768:
        X, Q = ds.lowrank_normalized(args.n, args.nq, dim=768, rank=32, seed=7712)
    else:
        X, Q = ds.sift_like(args.n, args.nq)
    if args.dtype == "uint8":
        X, Q = X.astype(np.uint8), Q.astype(np.uint8)
    index = flatnav.index.create(metric, DIM, args.n, 32, getattr(flatnav.data_type.DataType, args.dtype))
    index.set_num_threads(min(24, os.cpu_count()))
    t0 = time.time()
    index.add(X, 100, device=True)
    print("build %.1fs" % (time.time() - t0), flush=True)
    dev = hip.DeviceIndex.upload(np.asarray(index._raw_blob()), index._node_size_bytes, index._data_size_bytes, 32,
                                 args.n, args.dtype, metric, DIM)
    for o in args.opt:
        k, v = o.split("=")
        dev.set_option(k, int(v))
    L = hip.lib()
    L.fnv_debug_phase_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    for nq in sorted({1, 64, args.nq}):
        for kernel, opts in (("two heaps", {"sorted_beam": 0}), ("sorted beam", {"sorted_beam": 1})):
            for k, v in opts.items():
                dev.set_option(k, v)
            dev.search(Q[:nq], 10, args.ef)
            buf = (C.c_uint64 * 16)()
            L.fnv_debug_phase_cycles(dev._h, buf)  # reset after warm-up
            reps = 50 if nq == 1 else 1
            hops = 0; ms = 0.0; nd = 0.0
            for r in range(reps):
                _, _, st = dev.search(Q[r:r + nq], 10, args.ef, stats=True)
                ms += dev.last_kernel_ms(); hops += st["n_hops"].sum(); nd += st["n_dist"].sum()
            L.fnv_debug_phase_cycles(dev._h, buf)
            cyc = np.array(list(buf), dtype=np.float64)
            nqt = nq * reps
            print("\n%s, %d queries per launch, ef=%d: kernel %.3f ms per launch (with timing overhead), geometry %s" % (
                kernel, nq, args.ef, ms / reps, dev.launch_geometry()))
            print("%-18s %14s %12s %10s" % ("phase", "cycles/query", "cycles/hop", "share"))
            for name, c in zip(PHASES, cyc):
                if c:
                    print("%-18s %14.0f %12.1f %9.1f%%" % (name, c / nqt, c / hops, 100 * c / cyc.sum()))
            print("total cycles/query %.0f; hops/query %.1f; dist evals/query %.1f" % (cyc.sum() / nqt, hops / nqt, nd / nqt))


if __name__ == "__main__":
    main()
