#!/usr/bin/env python3
"""Developer tool: latency of small launches (1 ... 1024 queries) with a C-ABI option at two values, alternating in one process
on one 1M x 128 index; results compared bit for bit.
  python tools/dev/small_launch_ab.py visited_direct 0 1 [float32|uint8] [N]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np
import flatnav_amd as flatnav
from flatnav_amd import datasets as ds, hip

OPT, A, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
DT = sys.argv[4] if len(sys.argv) > 4 else "float32"
N = int(sys.argv[5]) if len(sys.argv) > 5 else 1_000_000
X, Q = ds.sift_like(N, 2048)
if DT == "uint8":
    X, Q = X.astype(np.uint8), Q.astype(np.uint8)
ix = flatnav.index.create("l2", 128, N, 32, index_data_type=getattr(flatnav.data_type.DataType, DT))
ix.set_num_threads(8)
ix.add(X, 100, device=True)
dev = hip.DeviceIndex(ctypes.c_void_p(ix.device_handle()), owned=False)
for ef in (50, 100, 200):
    for nq in (1, 4, 16, 64, 128, 256, 1024):
        ms = {A: [], B: []}
        same = True
        geom = {}
        for r in range(40):
            qq = Q[(r * nq) % 1024:(r * nq) % 1024 + nq]
            out = {}
            for v in ((A, B) if r % 2 == 0 else (B, A)):
                dev.set_option(OPT, v)
                out[v] = dev.search(qq, 10, ef, stats=True)
                ms[v].append(dev.last_kernel_ms())
                geom[v] = dev.launch_geometry()
            a, b = out[A], out[B]
            same &= np.array_equal(a[1], b[1]) and np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and all(np.array_equal(a[2][k], b[2][k]) for k in ("count", "n_dist", "n_hops"))
        pa, pb = np.percentile(ms[A][4:], 50), np.percentile(ms[B][4:], 50)
        print("%s ef=%3d batch %4d: kernel p50 %.3f ms at %s=%d (lds %d), %.3f ms at %d (lds %d) (%+.1f %%)  results %s"
              % (DT, ef, nq, pa, OPT, A, geom[A]["lds_bytes"], pb, B, geom[B]["lds_bytes"], (pb / pa - 1) * 100, "identical" if same else "DIFFER"), flush=True)
# the reference's per-query protocol (experiments/run-benchmark.py:66-82) through the Python surface, wall clock
for v in (A, B, A, B):
    dev.set_option(OPT, v)
    for ef in (50, 100):
        for q in Q[:50]:
            ix.search_single(q, 10, ef)
        lat = []
        for q in Q[:500]:
            t0 = time.perf_counter(); ix.search_single(q, 10, ef); lat.append(time.perf_counter() - t0)
        lat = np.array(lat) * 1e3
        print("%s=%d ef=%d search_single: p50 %.3f ms  p90 %.3f  p99 %.3f  (%.0f queries/s, per-query protocol)" % (OPT, v, ef, np.percentile(lat, 50), np.percentile(lat, 90), np.percentile(lat, 99), 1e3 / lat.mean()), flush=True)
