#!/usr/bin/env python3
"""Developer probe (round 2): replay reasons of the sorted-beam kernel per (K, ef), and two-heap vs sorted-beam timing."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from flatnav_amd import datasets as ds, hip
import flatnav_amd as flatnav
import ctypes

def build(metric, X, M=32, efc=100):
    ix = flatnav.index.create(metric, X.shape[1], X.shape[0], M)
    ix.set_num_threads(8)
    ix.add(X, efc, device=True)
    return ix, hip.DeviceIndex(ctypes.c_void_p(ix.device_handle()), owned=False)

what = sys.argv[1] if len(sys.argv) > 1 else "reasons"
if what == "reasons":
    kind = sys.argv[2] if len(sys.argv) > 2 else "sift"
    if kind == "sift":
        X, Q = ds.sift_like(1_000_000, 10000); metric = "l2"
    else:
        X, Q = ds.lowrank_normalized(1_000_000, 10000, dim=100, rank=24, seed=100); metric = "angular"
    ix, dev = build(metric, X)
    for K, ef in ((10, 32), (10, 52), (10, 100), (10, 200), (10, 400), (10, 800), (100, 400)):
        dev.set_option("sorted_beam", 1)
        d, l, st = dev.search(Q, K, ef, stats=True)
        r = dev.replayed_queries()
        dev.set_option("sorted_beam", 0)
        d0, l0, st0 = dev.search(Q, K, ef, stats=True)
        print(K, ef, r, "same", bool((l == l0).all() and (d == d0).all() and (st["n_dist"] == st0["n_dist"]).all()), "mean hops", st0["n_hops"].mean(), flush=True)
else:
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    kind = sys.argv[3] if len(sys.argv) > 3 else "sift"
    if kind == "sift":
        X, Q = ds.sift_like(N, 10000); metric = "l2"
    elif kind == "glove":
        X, Q = ds.lowrank_normalized(N, 10000, dim=100, rank=24, seed=100); metric = "angular"
    elif kind == "s3":
        X, Q = ds.lowrank_normalized(N, 10000, dim=768, rank=32, seed=7712); metric = "angular"
    else:
        X, Q = ds.randn(N, 10000, 128, seed=50); metric = "l2"
    ix, dev = build(metric, X)
    for ef in (32, 52, 64, 100, 200, 400, 800):
        modes = [("heaps", {"sorted_beam": 0})]
        for t in ((0, 50, 75, 100, 125, 150, 200) if ef <= 64 else (0, 12, 25, 50, 75)):
            modes.append(("sorted t%d" % t, {"sorted_beam": 1, "sorted_tail_exact_pct": t}))
        for mode, opts in modes:
            for k, v in opts.items(): dev.set_option(k, v)
            dev.search(Q, 10, ef)
            ts = []
            for _ in range(4):
                dev.search(Q, 10, ef); ts.append(dev.last_kernel_ms())
            g = dev.launch_geometry()
            print("ef=%d %-12s %.3f ms  %s bpc %d exact-reruns %s" % (ef, mode, min(ts), g["kernel"], g["blocks_per_cu"], dev.replayed_queries()["total"]), flush=True)
