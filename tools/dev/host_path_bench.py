#!/usr/bin/env python3
"""Developer tool: the host-buffer entry point (fnv_search_batch: queries and results in host memory, SURVEY.md 8d's
metric definition) against the device-pointer entry point, in one process.  Prints ms per 10 000-query call and
queries/s.  (Round 4 used it to measure two attempts at hiding the copies inside one call -- see DESIGN.md 5.)"""
import argparse, ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import flatnav_amd as flatnav
from flatnav_amd import hip

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c2")
ap.add_argument("--n", type=int, default=0)
ap.add_argument("--ef", type=int, default=52)
ap.add_argument("--dtype", default="float32")
ap.add_argument("--calls", type=int, default=12)
args = ap.parse_args()
cfg = dict(bench.CONFIGS[args.config]); N = args.n or cfg["n"]; DIM = cfg["dim"]; NQ, NB, K = 10_000, 6, 10
DT = cfg.get("dtype", args.dtype)
dev_t = torch.device("cuda", 0)
data = bench.Data(cfg, N, NQ * NB, torch, dev_t)
index = flatnav.index.create(distance_type=cfg["metric"], index_data_type=getattr(flatnav.data_type.DataType, DT), dim=DIM, dataset_size=N, max_edges_per_node=32)
index.set_num_threads(16); index.set_device(0)
for first, xh in data.chunks(5_000_000 if DIM <= 256 else 1_000_000):
    if DT == "uint8": xh = xh.astype(np.uint8)
    index.add(data=xh, ef_construction=100, labels=list(range(first, first + len(xh))), device=True)
dev = hip.DeviceIndex(ctypes.c_void_p(index.device_handle()), owned=False)
Q = data.queries()
if DT == "uint8": Q = Q.astype(np.uint8)
Q = np.ascontiguousarray(Q).reshape(NB, NQ, DIM)
dq = torch.from_numpy(Q).to(dev_t)
od = torch.empty((NQ, K), dtype=torch.float32, device=dev_t); ol = torch.empty((NQ, K), dtype=torch.int32, device=dev_t)
dev.tune(int(dq[0].data_ptr()), K, args.ef, 100, nq=NQ)

def device_calls(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        dev.search_device(dq[i % NB].data_ptr(), NQ, K, args.ef, 100, od.data_ptr(), ol.data_ptr())
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

def host_calls(n):
    ts = []
    for i in range(n):
        t0 = time.perf_counter(); r = dev.search(Q[i % NB], K, args.ef); ts.append((time.perf_counter() - t0) * 1e3)
    return ts, r

device_calls(3)
print("%s ef=%d device-resident: %.3f ms per call (%.0f q/s)" % (args.config, args.ef, (d := device_calls(args.calls)), NQ / d * 1e3), flush=True)
for rnd in range(3):
    host_calls(2)
    ts, r = host_calls(args.calls)
    print("host buffers: median %.3f ms  min %.3f  max %.3f  -> %.0f q/s (%.2f of the device-resident rate)"
          % (np.median(ts), min(ts), max(ts), NQ / np.median(ts) * 1e3, d / np.median(ts)), flush=True)
