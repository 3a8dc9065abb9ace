#!/usr/bin/env python3
"""Developer tool: do consecutive launches overlap (drain of one with the ramp of the next) when they go to two handles
(fnv_index_view) on two streams?  Bench protocol: 40 launches of 10 000 queries, rotating batches."""
import ctypes, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import flatnav_amd as flatnav
from flatnav_amd import datasets as ds, hip

which = sys.argv[1] if len(sys.argv) > 1 else "f32"
NB, NQ, K = 8, 10_000, 10
X, Q = ds.sift_like(1_000_000, NB * NQ)
dt = "float32"
if which == "u8":
    X, Q, dt = X.astype(np.uint8), Q.astype(np.uint8), "uint8"
index = flatnav.index.create("l2", 128, len(X), 32, getattr(flatnav.data_type.DataType, dt))
index.set_num_threads(16)
index.add(X, 100, device=True)
dev = hip.DeviceIndex(ctypes.c_void_p(index.device_handle()), owned=False)
views = [dev.view() for _ in range(3)]
dq = torch.from_numpy(Q.reshape(NB, NQ, -1)).cuda()
outs = [(torch.empty((NQ, K), dtype=torch.float32, device="cuda"), torch.empty((NQ, K), dtype=torch.int32, device="cuda")) for _ in range(4)]
streams = [torch.cuda.Stream() for _ in range(4)]
for ef in (52, 100):
    for h in [dev] + views:
        h.tune(int(dq[0].data_ptr()), K, ef, 100, nq=NQ)
    for nstreams in (1, 2, 3, 1, 2, 3):
        hs = ([dev] + views)[:nstreams]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        steps = 60
        for i in range(steps):
            j = i % nstreams
            hs[j].search_device(dq[i % NB].data_ptr(), NQ, K, ef, 100, outs[j][0].data_ptr(), outs[j][1].data_ptr(), stream=streams[j].cuda_stream)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print(which, "ef", ef, "streams", nstreams, "%.3f ms/launch  %.2f M queries/s" % (el / steps * 1e3, steps * NQ / el / 1e6), flush=True)
