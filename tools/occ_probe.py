#!/usr/bin/env python3
"""Developer probe: sorted-beam kernel timing per beam width for one library build (FLATNAV_HIP_LIB), one data kind."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, ctypes
from flatnav_amd import datasets as ds, hip
import flatnav_amd as flatnav

kind = sys.argv[1]; efs = [int(x) for x in sys.argv[2].split(",")]
opts = dict(kv.split("=") for kv in sys.argv[3:]) if len(sys.argv) > 3 else {}
N = 1_000_000
dtype = "float32"
if kind == "sift": X, Q = ds.sift_like(N, 10000); metric = "l2"
elif kind == "sift_u8": X, Q = ds.sift_like(N, 10000); X = X.astype(np.uint8); Q = Q.astype(np.uint8); metric = "l2"; dtype = "uint8"
elif kind == "glove": X, Q = ds.lowrank_normalized(N, 10000, dim=100, rank=24, seed=100); metric = "angular"
elif kind == "s3": N = 200_000; X, Q = ds.lowrank_normalized(N, 10000, dim=768, rank=32, seed=7712); metric = "angular"
else: X, Q = ds.randn(N, 10000, 128, seed=50); metric = "l2"
kw = {} if dtype == "float32" else {"index_data_type": getattr(flatnav.data_type.DataType, dtype)}
ix = flatnav.index.create(metric, X.shape[1], X.shape[0], 32, **kw)
ix.set_num_threads(8); ix.add(X, 100, device=True)
dev = hip.DeviceIndex(ctypes.c_void_p(ix.device_handle()), owned=False)
for ef in efs:
    for mode, o in (("heaps", {"sorted_beam": 0}), ("sorted", {"sorted_beam": 1, "sorted_tail_exact_pct": 0})):
        for k, v in {**o, **{k: int(v) for k, v in opts.items()}}.items(): dev.set_option(k, v)
        dev.search(Q, 10, ef)
        ts = []
        for _ in range(5):
            dev.search(Q, 10, ef); ts.append(dev.last_kernel_ms())
        g = dev.launch_geometry()
        print("%s ef=%d %-7s %.3f ms  %s bpc %d lds %d vis %d reruns %s" % (kind, ef, mode, min(ts), g["kernel"], g["blocks_per_cu"], g["lds_bytes"], g["visited_slots"], dev.replayed_queries()["total"]), flush=True)
