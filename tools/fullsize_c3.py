#!/usr/bin/env python3
"""Developer tool: BASELINE.json config C3 at its full size on one MI355X -- N x 768 float32, inner product
("angular" = 1 - <x,y> on normalised rows), M=32, 10 000 queries, K=10 -- now that the index can be built on the
GPU (index.add(..., device=True)).  Data: --kind lowrank (S3-like: rank-32 + 5 % noise, recall-qualified) or
randn (C3 as worded: recall is hopeless in 768 iid dimensions, roofline + parity only).  Generated in chunks with
torch on the GPU (seeded), so the 30 GB never exist twice on the host.
Reports: build time, QPS (kernel time by HIP events and host-buffer wall time), recall@10 against exact brute
force, algorithmic HBM GB/s, and GPU ids == CPU oracle ids on a sample (the oracle reads the same host blob)."""
import argparse, ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import flatnav_amd as flatnav
from flatnav_amd import datasets as ds, hip

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=10_000_000)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--kind", default="lowrank", choices=["lowrank", "randn"])
ap.add_argument("--metric", default="angular")
ap.add_argument("--efs", default="100,200,400")
ap.add_argument("--efc", type=int, default=100)
ap.add_argument("--chunk", type=int, default=1_000_000)
ap.add_argument("--oracle-sample", type=int, default=200)
ap.add_argument("--opt", action="append", default=[], help="name=value[,name=value] option sets to compare at every ef")
args = ap.parse_args()
NQ, K, M = 10000, 10, 32
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(7712)
W = torch.randn((32, args.dim), generator=g, device=dev) / (32 ** 0.5)


def gen(n):
    if args.kind == "lowrank":
        x = torch.randn((n, 32), generator=g, device=dev) @ W + 0.05 * torch.randn((n, args.dim), generator=g, device=dev)
    else:
        x = torch.randn((n, args.dim), generator=g, device=dev)
    if args.metric == "angular":
        x = x / x.norm(dim=1, keepdim=True)
    return x


threads = ds.effective_cpus() * 3 // 2
ix = flatnav.index.create(args.metric, args.dim, args.n, M, flatnav.data_type.DataType.float32, collect_stats=True)
ix.set_num_threads(threads)
chunks = []
t_gen = t_build = 0.0
for first in range(0, args.n, args.chunk):
    t0 = time.time(); x = gen(min(args.chunk, args.n - first)); xh = x.cpu().numpy(); t_gen += time.time() - t0
    chunks.append(x.half())  # kept in HBM (fp16) only for the brute-force ground truth
    t0 = time.time(); ix.add(xh, args.efc, labels=list(range(first, first + len(xh))), device=True); t_build += time.time() - t0
    print("  %d nodes: build %.1fs so far (data generation + D2H %.1fs)" % (first + len(xh), t_build, t_gen), flush=True)
    del xh, x
Q = gen(NQ)
Qh = Q.cpu().numpy()
print("index: %d x %d %s %s, M=%d, efc=%d: device build %.1f s (%d host threads for the bootstrap)" %
      (args.n, args.dim, args.kind, args.metric, M, args.efc, t_build, threads), flush=True)

# exact ground truth for 1000 queries (fp32 scores of the fp16-rounded copy are good enough to rank true neighbours:
# recomputed exactly below on the top 100 per chunk)
NG = 1000
best_s = torch.full((NG, K), -1e30, device=dev); best_i = torch.zeros((NG, K), dtype=torch.int64, device=dev)
Xfull = None
for ci, xc in enumerate(chunks):
    xf = xc.float()
    s = Q[:NG] @ xf.T if args.metric == "angular" else -(torch.cdist(Q[:NG], xf) ** 2)
    ts, ti = s.topk(K, dim=1)
    cs = torch.cat([best_s, ts], 1); cidx = torch.cat([best_i, ti + ci * args.chunk], 1)
    o = cs.topk(K, dim=1).indices
    best_s = cs.gather(1, o); best_i = cidx.gather(1, o)
    del xf, s
gt = best_i.cpu().numpy()
del chunks
torch.cuda.empty_cache()

d = hip.DeviceIndex(ctypes.c_void_p(ix.device_handle()))
esz, nscan = 4, -(-args.n // max(1, args.n // 100))
try:
    for ef, optset in [(int(e), o) for e in args.efs.split(",") for o in (args.opt or [""])]:
        applied = [kv.split("=") for kv in optset.split(",") if kv]
        for k, v in applied: d.set_option(k, int(v))
        if optset: print("options:", optset, flush=True)
        d.search(Qh, K, ef)
        t0 = time.perf_counter(); dd, ll, st = d.search(Qh, K, ef, stats=True); wall = time.perf_counter() - t0
        ms = d.last_kernel_ms()
        byts = float(((nscan + st["n_dist"]) * args.dim * esz + st["n_hops"] * M * 4 + K * 4).sum())
        print("ef=%3d: %8.0f QPS kernel (%.2f ms), %8.0f QPS host buffers  recall@10 %.4f  evals/q %.0f  %.2f TB/s algorithmic "
              "(%.2f of 8)  %s" % (ef, NQ / ms * 1e3, ms, NQ / wall, ds.recall_at_k(ll[:NG], gt), st["n_dist"].mean(),
                                  byts / ms / 1e9, byts / ms / 1e9 / 8, d.launch_geometry()), flush=True)
        defaults = {"cand_factor": 2, "visited_floor": 2048, "visited_factor": 27, "occupancy_target": 13}
        for k, v in applied: d.set_option(k, defaults.get(k, 0))
    if args.oracle_sample:
        from oracle import oracle as orc
        o = orc.OracleIndex.from_blob(args.metric, "float32", args.dim, args.n, args.n, M, np.asarray(ix._raw_blob()))
        ef = int(args.efs.split(",")[-1])
        t0 = time.time(); od, ol = o.search(Qh[:args.oracle_sample], K, ef, threads=ds.effective_cpus()); t = time.time() - t0
        _, ll = d.search(Qh[:args.oracle_sample], K, ef)
        print("CPU oracle on the same blob, ef=%d: %.0f QPS on %d threads; GPU ids == CPU ids on %.1f%% of %d queries" %
              (ef, args.oracle_sample / t, ds.effective_cpus(), 100.0 * (ol == ll).all(axis=1).mean(), args.oracle_sample), flush=True)
finally:
    d._h = None  # the handle belongs to ix
