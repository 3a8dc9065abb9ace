#!/usr/bin/env python3
"""Developer tool: does fnv_tune's measured LDS layout beat the rules' layout under the bench protocol (20 launches over
rotating batches, HIP events)?  A/B in one process, alternating."""
import ctypes, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import flatnav_amd as flatnav
from flatnav_amd import datasets as ds, hip

which = sys.argv[1] if len(sys.argv) > 1 else "c4"
NB, NQ, K = 8, 10_000, 10
if which == "c4":
    X, Q = ds.lowrank_normalized(1_183_514, NB * NQ, dim=100, rank=24, seed=100); metric, dt = "angular", "float32"
elif which == "u8":
    X, Q = ds.sift_like(1_000_000, NB * NQ); X, Q = X.astype(np.uint8), Q.astype(np.uint8); metric, dt = "l2", "uint8"
else:
    X, Q = ds.sift_like(1_000_000, NB * NQ); metric, dt = "l2", "float32"
index = flatnav.index.create(metric, X.shape[1], len(X), 32, getattr(flatnav.data_type.DataType, dt))
index.set_num_threads(16)
index.add(X, 100, device=True)
dev = hip.DeviceIndex(ctypes.c_void_p(index.device_handle()), owned=False)
dq = torch.from_numpy(Q.reshape(NB, NQ, -1)).cuda()
od = torch.empty((NQ, K), dtype=torch.float32, device="cuda"); ol = torch.empty((NQ, K), dtype=torch.int32, device="cuda")
stream = torch.cuda.current_stream()

def timed(ef, steps=24):
    evs = []
    for i in range(steps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        dev.search_device(dq[i % NB].data_ptr(), NQ, K, ef, 100, od.data_ptr(), ol.data_ptr(), stream=stream.cuda_stream)
        b.record(stream); evs.append((a, b))
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in evs]))

for ef in [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "52,100,110,200,400".split(","))]:
    res = {}
    for rnd in range(3):
        for lay in (0, 1):
            dev.set_option("tune_layout", lay)
            dev.tune(int(dq[0].data_ptr()), K, ef, 100, nq=NQ)
            timed(ef, 4)
            ms = timed(ef)
            g = dev.launch_geometry()
            res.setdefault(lay, []).append((round(ms, 4), g["blocks_per_cu"], g["visited_slots"], g["cand_slots"], dev.launch_info()["variant"]))
    print(which, "ef", ef, "rules:", res[0], " tuned:", res[1], flush=True)
