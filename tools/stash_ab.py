#!/usr/bin/env python3
"""Developer tool: batch time per visited-table size (option visited_slots) -- run once per library build
(FLATNAV_HIP_LIB) to see what the overflow stash buys at small tables.  usage: stash_ab.py c2|u8|c4 ef[,ef..]"""
import ctypes, sys, os
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

dev.set_option("tune_layout", 0)
tag = os.path.basename(os.environ.get("FLATNAV_HIP_LIB", "default"))
for ef in [int(x) for x in sys.argv[2].split(",")]:
    for slots in (0, 512, 768, 1024, 1536, 2048, 3072, 4096, 6144):
        dev.set_option("visited_slots", slots)
        try:
            dev.tune(int(dq[0].data_ptr()), K, ef, 100, nq=NQ)
        except Exception as e:
            print(tag, which, "ef", ef, "slots", slots, "skip", e, flush=True); continue
        timed(ef, 4)
        ms = min(timed(ef), timed(ef))
        g = dev.launch_geometry()
        print(tag, which, "ef", ef, "slots", slots, "->", g["visited_slots"], "ms %.4f" % ms, "bpc", g["blocks_per_cu"],
              g["kernel"], dev.launch_info()["variant"], flush=True)
