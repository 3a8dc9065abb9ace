#!/usr/bin/env python3
"""Developer tool: does relabelling the graph with the reference's reordering strategies
(gorder / rcm) change GPU search throughput?  Builds the bench index, measures device-resident QPS / recall /
evals per query at a few ef, applies index.reorder([...]) and measures again.  A relabelling keeps the graph
isomorphic but moves the entry-scan sample, so results are statistically -- not bitwise -- the same."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import flatnav_amd as flatnav
from flatnav_amd import datasets as ds, hip

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--efs", default="50,100")
ap.add_argument("--strategies", default="gorder;rcm;gorder,rcm")
args = ap.parse_args()
NQ, K, M = 10000, 10, 32
X, Q = ds.sift_like(args.n, NQ)
gt = ds.exact_topk_l2(X, Q[:1000], K)
ix = flatnav.index.create("l2", 128, args.n, M, flatnav.data_type.DataType.float32)
ix.set_num_threads(ds.effective_cpus() * 3 // 2)
t0 = time.time(); ix.add(X, 100); print("build %.1fs" % (time.time() - t0), flush=True)
labels_of = None


def measure(tag):
    dev = hip.DeviceIndex.upload(np.asarray(ix._raw_blob()), ix._node_size_bytes, ix._data_size_bytes, M, args.n,
                                 "float32", "l2", 128)
    dq = torch.from_numpy(Q).cuda(); dd = torch.empty((NQ, K), dtype=torch.float32, device="cuda")
    dl = torch.empty((NQ, K), dtype=torch.int32, device="cuda")
    nd = torch.zeros(NQ, dtype=torch.int64, device="cuda"); nh = torch.zeros(NQ, dtype=torch.int64, device="cuda")
    for ef in [int(e) for e in args.efs.split(",")]:
        for _ in range(2):
            dev.search_device(dq.data_ptr(), NQ, K, ef, 100, dd.data_ptr(), dl.data_ptr(), 0, nd.data_ptr(), nh.data_ptr())
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            dev.search_device(dq.data_ptr(), NQ, K, ef, 100, dd.data_ptr(), dl.data_ptr(), 0, nd.data_ptr(), nh.data_ptr())
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        rec = ds.recall_at_k(dl.cpu().numpy()[:1000], gt)
        print("%-14s ef=%3d: %9.0f QPS  recall@10 %.4f  evals/q %.0f  hops/q %.1f" %
              (tag, ef, NQ / dt, rec, float(nd.float().mean()), float(nh.float().mean())), flush=True)
    dev.close()


measure("as built")
for strat in args.strategies.split(";"):
    t0 = time.time(); ix.reorder(strat.split(",")); t = time.time() - t0
    print("reorder %s: %.1fs (cumulative on the previous order)" % (strat, t), flush=True)
    measure(strat)
