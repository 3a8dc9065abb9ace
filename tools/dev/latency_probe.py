#!/usr/bin/env python3
"""Developer tool: latency of small batches on the 1M x 128 bench index -- the reference's per-query protocol
(experiments/run-benchmark.py:66-82: one search call per query) and batches of 1..1024, wall time vs kernel time."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np
import flatnav_amd as flatnav
from flatnav_amd import datasets as ds, hip

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
X, Q = ds.sift_like(N, 2048)
ix = flatnav.index.create("l2", 128, N, 32)
ix.set_num_threads(8)
ix.add(X, 100, device=True)
dev = hip.DeviceIndex(ctypes.c_void_p(ix.device_handle()), owned=False)
for kv in sys.argv[2:]:  # library options, e.g. beam_registers=0
    k, v = kv.split("="); dev.set_option(k, int(v))
for ef in (50, 100, 200):
    for q in Q[:50]:
        ix.search_single(q, 10, ef)
    lat = []
    for q in Q[:500]:
        t0 = time.perf_counter(); ix.search_single(q, 10, ef); lat.append(time.perf_counter() - t0)
    lat = np.array(lat) * 1e3
    kms = []
    for q in Q[:100]:
        dev.search(q[None, :], 10, ef); kms.append(dev.last_kernel_ms())
    print("ef=%d search_single: p50 %.3f ms  p90 %.3f  p99 %.3f  (%.0f QPS per-query protocol); kernel alone p50 %.3f ms  %s"
          % (ef, np.percentile(lat, 50), np.percentile(lat, 90), np.percentile(lat, 99), 1e3 / lat.mean(), np.percentile(kms, 50),
             dev.launch_geometry()["kernel"]), flush=True)
    for nq in (1, 4, 16, 64, 256, 1024):
        ts, ks = [], []
        for r in range(20):
            qq = Q[(r * nq) % 1024:(r * nq) % 1024 + nq]
            t0 = time.perf_counter(); dev.search(qq, 10, ef); ts.append(time.perf_counter() - t0); ks.append(dev.last_kernel_ms())
        print("   batch %4d: wall p50 %.3f ms, kernel p50 %.3f ms -> %.0f QPS" % (nq, np.percentile(ts, 50) * 1e3, np.percentile(ks, 50), nq / np.percentile(ts, 50)), flush=True)
    # the reference's per-query protocol from SEVERAL caller threads on the one handle (round 4: concurrent callers run on
    # the handle's hidden lanes -- up to 8 single queries in flight -- instead of taking turns)
    import threading
    for T in (1, 2, 4, 8, 16):
        per = 200
        def worker(t):
            for i in range(per):
                dev.search(Q[(t * per + i) % 2048][None, :], 10, ef)
        worker(0)
        th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        wall = time.perf_counter() - t0
        print("   %2d caller threads x %d single queries: %.0f queries/s (%.3f ms per query per thread)" % (T, per, T * per / wall, wall / per * 1e3), flush=True)
