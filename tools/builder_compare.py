#!/usr/bin/env python3
"""Developer tool: host builder vs device builder on the SAME data at a size where the question matters (VERDICT r1:
S3 at 10M needed ef=800 for recall 0.95 on a device-built graph -- is that N, or the batched insertion?).
Builds an N x 768 S3 (low-rank unit vectors, inner product) index twice and reports recall@10 per ef for both."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
import flatnav_amd as flatnav
from flatnav_amd import datasets as ds

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
DIM, M, NQ, K = 768, 32, 2000, 10
X, Q = ds.lowrank_normalized(N, NQ, dim=DIM, rank=32, seed=7712)
xt = torch.from_numpy(X).cuda(); qt = torch.from_numpy(Q).cuda()
gt = torch.cat([torch.topk(qt[s:s + 500] @ xt.T, K, dim=1).indices for s in range(0, NQ, 500)]).cpu().numpy()
del xt, qt; torch.cuda.empty_cache()
for name, kw in (("device builder (batches <= 32768)", dict(device=True)),
                 ("device builder (batches <= 4096)", dict(device=True, device_max_batch=4096)),
                 ("host builder (%d threads)" % (ds.effective_cpus() * 3 // 2), dict())):
    ix = flatnav.index.create("angular", DIM, N, M)
    ix.set_num_threads(ds.effective_cpus() * 3 // 2)
    t0 = time.time(); ix.add(X, 100, **kw); t = time.time() - t0
    row = []
    for ef in (100, 200, 400, 800):
        _, l = ix.search(Q, K, ef)
        row.append("ef=%d %.4f" % (ef, ds.recall_at_k(l, gt)))
    blob = np.asarray(ix._raw_blob()).reshape(N, ix._node_size_bytes)
    links = blob[:, DIM * 4:DIM * 4 + 4 * M].copy().view(np.uint32)
    deg = (links != np.arange(N, dtype=np.uint32)[:, None]).sum(1)
    print("%-36s build %.1fs  recall@10: %s  mean out-degree %.2f" % (name, t, "  ".join(row), deg.mean()), flush=True)
    del ix
