#!/usr/bin/env python3
"""Developer tool: the other BASELINE.json configurations at reduced N (they are parity-test cases, not the
bench line): C3-like 768-d inner product, C4-like 100-d angular, uint8 SIFT, randn L2.  For each: build with the
product's host builder, search on the GPU (device-resident buffers), compare ids with the CPU oracle on a
sample, report QPS / recall@10 / algorithmic GB/s."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import flatnav_amd as flatnav
from flatnav_amd import datasets as ds, hip
from oracle import oracle as orc

ap = argparse.ArgumentParser(); ap.add_argument("--only", default=""); args = ap.parse_args()
NQ, K = 10000, 10
CASES = [
    # name, metric, dtype, generator -> (X, Q), efs
    ("C3-like lowrank 200k x 768 IP (S3)", "angular", "float32", lambda: ds.lowrank_normalized(200_000, NQ, 768, 32, 7712), [100, 200]),
    ("C3 as worded: randn-normalised 200k x 768 IP", "angular", "float32", lambda: ds.randn(200_000, NQ, 768, 768, normalize=True), [200]),
    ("C4-like lowrank 1M x 100 angular", "angular", "float32", lambda: ds.lowrank_normalized(1_000_000, NQ, 100, 24, 100), [50, 100, 200]),
    ("SIFT stand-in as uint8 1M x 128 L2", "l2", "uint8", lambda: tuple(a.astype(np.uint8) for a in ds.sift_like(1_000_000, NQ)), [60, 100]),
    ("C5-like randn 1M x 128 L2", "l2", "float32", lambda: ds.randn(1_000_000, NQ, 128, 50), [100, 400]),
]
threads = ds.effective_cpus() * 3 // 2
for name, metric, dt, gen, efs in CASES:
    if args.only and args.only not in name: continue
    t0 = time.time(); X, Q = gen(); N, dim = X.shape
    ix = flatnav.index.create(metric, dim, N, 32, getattr(flatnav.data_type.DataType, dt)); ix.set_num_threads(threads)
    ix.add(X, 100); tb = time.time() - t0
    dev = hip.DeviceIndex.upload(np.asarray(ix._raw_blob()), ix._node_size_bytes, ix._data_size_bytes, 32, N, dt, metric, dim)
    o = orc.OracleIndex.from_blob(metric, dt, dim, N, N, 32, np.asarray(ix._raw_blob()))
    gt = (ds.exact_topk_l2 if metric == "l2" else ds.exact_topk_ip)(X.astype(np.float32), Q[:300].astype(np.float32), K)
    dq = torch.from_numpy(Q).cuda(); dd = torch.empty((NQ, K), dtype=torch.float32, device="cuda")
    dl = torch.empty((NQ, K), dtype=torch.int32, device="cuda"); nd = torch.zeros(NQ, dtype=torch.int64, device="cuda"); nh = torch.zeros(NQ, dtype=torch.int64, device="cuda")
    print("%s  (build %.0fs, %d threads)" % (name, tb, threads), flush=True)
    for ef in efs:
        for _ in range(2):
            dev.search_device(dq.data_ptr(), NQ, K, ef, 100, dd.data_ptr(), dl.data_ptr(), 0, nd.data_ptr(), nh.data_ptr())
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            dev.search_device(dq.data_ptr(), NQ, K, ef, 100, dd.data_ptr(), dl.data_ptr(), 0, nd.data_ptr(), nh.data_ptr())
        torch.cuda.synchronize(); dt_s = (time.perf_counter() - t0) / 3
        lab = dl.cpu().numpy(); rec = ds.recall_at_k(lab[:300], gt)
        od, ol = o.search(Q[:500], K, ef, threads=ds.effective_cpus())
        same = float((ol == lab[:500]).all(axis=1).mean())
        esz = X.dtype.itemsize; nscan = -(-N // max(1, N // 100))
        byts = float(((nscan + nd.cpu().numpy()) * dim * esz + nh.cpu().numpy() * 32 * 4 + K * 4).sum())
        print("   ef=%3d: %8.0f QPS  recall@10 %.3f  %.2f TB/s algorithmic (%.2f of 8)  evals/q %.0f  GPU ids == CPU ids on %.1f%% of 500  %s"
              % (ef, NQ / dt_s, rec, byts / dt_s / 1e12, byts / dt_s / 8e12, float(nd.float().mean()), same * 100, dev.launch_geometry()), flush=True)
    del dev, o, ix
