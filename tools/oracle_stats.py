#!/usr/bin/env python3
"""Developer tool: distribution of per-query beam-state sizes (CPU oracle) on the bench workload,
used to size the kernel's LDS structures (candidate heap, visited table)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import flatnav_amd as flatnav
from flatnav_amd import datasets as ds
from oracle import oracle as orc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
X, Q = ds.sift_like(n, 4000)
index = flatnav.index.create("l2", 128, n, 32)
index.set_num_threads(min(24, os.cpu_count()))
t0 = time.time(); index.add(X, 100); print("build %.1fs" % (time.time() - t0), flush=True)
o = orc.OracleIndex.from_blob("l2", "float32", 128, n, n, 32, np.asarray(index._raw_blob()))
for ef in (50, 100, 200, 400):
    _, _, st = o.search(Q, 10, ef, threads=64, stats=True)
    pct = lambda a: [int(np.percentile(a, p)) for p in (50, 90, 99, 99.9, 100)]
    print("ef=%d max_cand p50/90/99/99.9/max %s  n_dist %s  n_admit %s  hops %s" % (
        ef, pct(st["max_cand"]), pct(st["n_dist"]), pct(st["n_admit"]), pct(st["n_hops"])), flush=True)
