#!/usr/bin/env python3
"""Developer tool: merged-beam kernel against the two-heap kernel on an index with more than 2^24 nodes (node ids and
visited tags beyond 24 bits come into play naturally, not by option)."""
import sys, os, ctypes, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import flatnav_amd as flatnav
from flatnav_amd import hip
N, dim, M = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000, 16, 16
rng = np.random.default_rng(5)
X = rng.integers(0, 40, (N, dim)).astype(np.uint8); Q = rng.integers(0, 40, (2000, dim)).astype(np.uint8)
ix = flatnav.index.create("l2", dim, N, M, index_data_type=flatnav.data_type.DataType.uint8)
ix.set_num_threads(8)
t0 = time.time(); ix.add(X, 40, device=True); print("build %.1fs" % (time.time() - t0), flush=True)
dev = hip.DeviceIndex(ctypes.c_void_p(ix.device_handle()), owned=False)
for K, ef in ((10, 40), (10, 150), (50, 400)):
    dev.set_option("sorted_beam", 0); w = dev.search(Q, K, ef, stats=True)
    for regs in (1, 0):
        dev.set_option("sorted_beam", 1); dev.set_option("beam_registers", regs)
        g = dev.search(Q, K, ef, stats=True)
        same = np.array_equal(w[1], g[1]) and np.array_equal(w[0].view(np.uint32), g[0].view(np.uint32)) and np.array_equal(w[2]["n_dist"], g[2]["n_dist"]) and np.array_equal(w[2]["n_hops"], g[2]["n_hops"])
        print("K", K, "ef", ef, dev.launch_geometry(), "reruns", dev.replayed_queries()["total"], "identical", same, "max id", int(g[1].max()), flush=True)
        assert same
