#!/usr/bin/env python3
"""Developer tool: ONE host call with a big batch (20 000 ... 80 000 queries) -- plain fnv_search_batch against the same rows
sharded over G handles of the same index driven by G host threads (fnv_search_batch_multi with the handle listed G times:
shard g > 0 runs on the handle's hidden lanes)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import flatnav_amd as flatnav
from flatnav_amd import hip

cfg = dict(bench.CONFIGS["c2"]); N, DIM, K, EF = cfg["n"], cfg["dim"], 10, 52
data = bench.Data(cfg, N, 80_000, torch, torch.device("cuda", 0))
index = flatnav.index.create(distance_type=cfg["metric"], index_data_type=flatnav.data_type.DataType.float32, dim=DIM, dataset_size=N, max_edges_per_node=32)
index.set_num_threads(16); index.set_device(0)
index.add(data=data.X, ef_construction=100, device=True)
dev = hip.DeviceIndex(ctypes.c_void_p(index.device_handle()), owned=False)
Q = np.ascontiguousarray(data.queries())
dev.tune(Q[:10000], K, EF)
for nq in (20_000, 40_000, 80_000):
    ref = dev.search(Q[:nq], K, EF)
    for G in (1, 2, 3, 4):
        hs = [dev] * G
        hip.search_multi(hs, Q[:nq], K, EF)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); r = hip.search_multi(hs, Q[:nq], K, EF); ts.append(time.perf_counter() - t0)
        assert np.array_equal(r[1], ref[1])
        print("nq=%d over %d shard(s) of one handle: %.3f ms -> %.2f M queries/s" % (nq, G, np.median(ts) * 1e3, nq / np.median(ts) / 1e6), flush=True)
