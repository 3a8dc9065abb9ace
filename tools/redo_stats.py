#!/usr/bin/env python3
"""Developer tool: how many queries does the register-beam kernel hand to the exact kernel, and why, on the bench
data set (integer-valued float32), its uint8 form, and non-integer float data; plus timing fast+replay vs exact."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import flatnav_amd as flatnav
from flatnav_amd import datasets as ds, hip
N, NQ, K, M = int(os.environ.get("N", 1_000_000)), 10000, 10, 32
ONLY = os.environ.get("ONLY", "")
def run(name, metric, dt, X, Q, efs):
    if ONLY and ONLY not in name: return
    ix = flatnav.index.create(metric, X.shape[1], N, M, getattr(flatnav.data_type.DataType, dt))
    ix.set_num_threads(ds.effective_cpus() * 3 // 2); ix.add(X, 100, device=True)
    ix.search(Q[:10], K, 50)
    import ctypes
    d = hip.DeviceIndex(ctypes.c_void_p(ix.device_handle()))
    dq = torch.from_numpy(Q).cuda(); dd = torch.empty((NQ, K), dtype=torch.float32, device="cuda"); dl = torch.empty((NQ, K), dtype=torch.int32, device="cuda")
    try:
        for ef in efs:
            res = {}
            for exact in (1, 0):
                d.set_option("sorted_beam", 1 - exact)
                for _ in range(2): d.search_device(dq.data_ptr(), NQ, K, ef, 100, dd.data_ptr(), dl.data_ptr())
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(5): d.search_device(dq.data_ptr(), NQ, K, ef, 100, dd.data_ptr(), dl.data_ptr())
                torch.cuda.synchronize(); res[exact] = ((time.perf_counter() - t0) / 5, dl.cpu().numpy().copy(), dd.cpu().numpy().copy())
            r = d.replayed_queries()
            same = np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
            print("%-28s ef=%2d: exact %.3f ms, register-beam + replay %.3f ms (x%.2f)  replayed %s  identical=%s" %
                  (name, ef, res[1][0] * 1e3, res[0][0] * 1e3, res[1][0] / res[0][0], r, same), flush=True)
    finally:
        d._h = None
X, Q = ds.sift_like(N, NQ)
run("sift-like f32 (integers)", "l2", "float32", X, Q, (50, 64))
run("sift-like uint8", "l2", "uint8", X.astype(np.uint8), Q.astype(np.uint8), (50, 64))
X, Q = ds.lowrank_normalized(N, NQ, 100, 24, 100)
run("lowrank 100-d angular f32", "angular", "float32", X, Q, (50, 64))
X, Q = ds.randn(N, NQ, 128, 50)
run("randn 128-d l2 f32", "l2", "float32", X, Q, (50, 64))
