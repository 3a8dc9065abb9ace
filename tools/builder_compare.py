#!/usr/bin/env python3
"""Developer tool: host builder vs device builder on the SAME data at the sizes where the question matters (VERDICT r1 / r2:
S3 at 10M needs ef=800 for recall 0.95 on a device-built graph, 50M x 128 randn returns recall 0.048 at ef=100 -- is that
the size / the data, or the batched insertion?).  Builds the index with each builder and reports recall@10 per ef against
exact ground truth computed on the GPU, plus graph statistics.

  python tools/builder_compare.py lowrank768 10000000      # S3 low-rank unit vectors, inner product
  python tools/builder_compare.py randn128 10000000        # isotropic Gaussian, L2 (configuration C5's data)
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
import flatnav_amd as flatnav
from flatnav_amd import datasets as ds

kind = sys.argv[1] if len(sys.argv) > 1 else "lowrank768"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
NQ = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000
M, K = 32, 10
t0 = time.time()
if kind == "lowrank768":
    DIM, metric, efs = 768, "angular", (100, 200, 400, 800)
    X, Q = ds.lowrank_normalized(N, NQ, dim=DIM, rank=32, seed=7712)
else:
    DIM, metric, efs = 128, "l2", (100, 200, 400, 800)
    X, Q = ds.randn(N, NQ, DIM, seed=50)
print("%s N=%d: data %.0fs" % (kind, N, time.time() - t0), flush=True)
qt = torch.from_numpy(Q).cuda()
best_s = torch.full((NQ, K), float("inf"), device="cuda"); best_i = torch.zeros((NQ, K), dtype=torch.int64, device="cuda")
for s in range(0, N, 500_000):  # exact top-K: scores per block of rows, merged
    xb = torch.from_numpy(X[s:s + 500_000]).cuda()
    for q0 in range(0, NQ, 2500):
        qq = qt[q0:q0 + 2500]
        sc = -(qq @ xb.T) if metric == "angular" else ((xb * xb).sum(1)[None, :] - 2.0 * (qq @ xb.T))
        cs, ci = torch.topk(sc, K, dim=1, largest=False)
        alls = torch.cat([best_s[q0:q0 + 2500], cs], 1); alli = torch.cat([best_i[q0:q0 + 2500], ci + s], 1)
        o = torch.topk(alls, K, dim=1, largest=False).indices
        best_s[q0:q0 + 2500] = alls.gather(1, o); best_i[q0:q0 + 2500] = alli.gather(1, o)
    del xb
gt = best_i.cpu().numpy()
del qt, best_s, best_i; torch.cuda.empty_cache()
threads = ds.effective_cpus() * 3 // 2
for name, kw in (("device builder (batches <= 32768)", dict(device=True)),
                 ("host builder (%d threads)" % threads, dict())):
    ix = flatnav.index.create(metric, DIM, N, M)
    ix.set_num_threads(threads)
    t0 = time.time()
    for s in range(0, N, 1_000_000):
        ix.add(X[s:s + 1_000_000], 100, labels=list(range(s, min(N, s + 1_000_000))), **kw)
    t = time.time() - t0
    row = []
    for ef in efs:
        _, l = ix.search(Q, K, ef)
        row.append("ef=%d %.4f" % (ef, ds.recall_at_k(l, gt)))
    blob = np.asarray(ix._raw_blob()).reshape(N, ix._node_size_bytes)
    links = blob[:, DIM * 4:DIM * 4 + 4 * M].copy().view(np.uint32)
    deg = (links != np.arange(N, dtype=np.uint32)[:, None]).sum(1)
    indeg = np.bincount(links[links != np.arange(N, dtype=np.uint32)[:, None]].ravel(), minlength=N)
    print("%-36s build %.1fs  recall@10 (%d queries): %s  mean out-degree %.2f  nodes without in-links %.3f%%"
          % (name, t, NQ, "  ".join(row), deg.mean(), 100.0 * (indeg == 0).mean()), flush=True)
    del ix, blob, links
