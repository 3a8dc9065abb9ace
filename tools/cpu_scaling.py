#!/usr/bin/env python3
"""Developer tool: thread scaling of the CPU oracle's batched search on the bench workload."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import flatnav_amd as flatnav
from flatnav_amd import datasets as ds
from oracle import oracle as orc
n = 1_000_000
X, Q = ds.sift_like(n, 10000)
index = flatnav.index.create("l2", 128, n, 32); index.set_num_threads(24); index.add(X, 100)
o = orc.OracleIndex.from_blob("l2", "float32", 128, n, n, 32, np.asarray(index._raw_blob()))
for ref in (False, True):
    o.use_reference_distance(ref)
    for th in (1, 16, 64, 128, 256):
        q = Q[:2000] if th == 1 else np.concatenate([Q] * 4)
        o.search(q[:500], 10, 60, threads=th)
        t0 = time.perf_counter(); o.search(q, 10, 60, threads=th); dt = time.perf_counter() - t0
        print("ref_dist=%s threads=%3d: %.0f QPS (%d queries, %.2fs)" % (ref, th, len(q) / dt, len(q), dt), flush=True)
