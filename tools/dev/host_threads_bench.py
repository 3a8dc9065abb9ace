#!/usr/bin/env python3
"""Developer tool: the host-buffer entry point from 1 ... T caller threads on ONE handle (10 000-query batches from pageable
numpy arrays; each caller is synchronous, concurrent callers run on the handle's hidden lanes).  Prints queries/s."""
import argparse, ctypes, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import flatnav_amd as flatnav
from flatnav_amd import hip

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c2")
ap.add_argument("--ef", type=int, default=52)
ap.add_argument("--threads", default="1,2,3,4")
args = ap.parse_args()
cfg = dict(bench.CONFIGS[args.config]); N = cfg["n"]; DIM = cfg["dim"]; NQ, NB, K = 10_000, 8, 10
DT = cfg.get("dtype", "float32")
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
dev.tune(Q[0], K, args.ef)
for T in [int(x) for x in args.threads.split(",")]:
    per = 8
    def worker(t):
        for i in range(per):
            dev.search(Q[(t + i) % NB], K, args.ef)
    for rnd in range(3):
        th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        wall = time.perf_counter() - t0
        if rnd: print("%s ef=%d: %d caller threads: %.0f queries/s" % (args.config, args.ef, T, T * per * NQ / wall), flush=True)
