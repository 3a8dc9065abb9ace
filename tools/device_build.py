#!/usr/bin/env python3
"""Developer tool: host builder vs device-assisted builder (index.add(..., device=True)) on the
bench data set: build time, then recall@10 / evals per query / QPS of the resulting graphs at a few ef."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import flatnav_amd as flatnav
from flatnav_amd import datasets as ds

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--dim", type=int, default=128)
ap.add_argument("--efs", default="50,100,200")
ap.add_argument("--efc", type=int, default=100)
ap.add_argument("--max-batch", default="32768")
ap.add_argument("--skip-host", action="store_true")
ap.add_argument("--wiring", default="both", choices=["both", "device", "host"])
args = ap.parse_args()
NQ, K, M = 10000, 10, 32
X, Q = ds.sift_like(args.n, NQ) if args.dim == 128 else ds.lowrank_normalized(args.n, NQ, args.dim, 32, 7712)
metric = "l2" if args.dim == 128 else "angular"
gt = (ds.exact_topk_l2 if metric == "l2" else ds.exact_topk_ip)(X, Q[:1000], K)
threads = ds.effective_cpus() * 3 // 2


def report(tag, ix):
    for ef in [int(e) for e in args.efs.split(",")]:
        ix.search(Q, K, ef)
        ix.get_query_distance_computations()
        t0 = time.perf_counter(); d, l = ix.search(Q, K, ef); dt = time.perf_counter() - t0
        evals = ix.get_query_distance_computations() / NQ
        print("%-40s ef=%3d: recall@10 %.4f  evals/q %.0f  %.0f QPS (host buffers)" %
              (tag, ef, ds.recall_at_k(l[:1000], gt), evals, NQ / dt), flush=True)
    tab = ix.get_graph_outdegree_table()
    print("%-40s mean out-degree %.2f" % (tag, float(np.mean([len(r) for r in tab[:100000]]))), flush=True)


if not args.skip_host:
    ix = flatnav.index.create(metric, args.dim, args.n, M, flatnav.data_type.DataType.float32, collect_stats=True)
    ix.set_num_threads(threads)
    t0 = time.time(); ix.add(X, args.efc); print("host builder (%d threads): %.1fs" % (threads, time.time() - t0), flush=True)
    report("host", ix); del ix
for mb in [int(b) for b in args.max_batch.split(",")]:
  for wiring in {"both": (True, False), "device": (True,), "host": (False,)}[args.wiring]:
    ix = flatnav.index.create(metric, args.dim, args.n, M, flatnav.data_type.DataType.float32, collect_stats=True)
    ix.set_num_threads(threads)
    t0 = time.time(); ix.add(X, args.efc, device=True, device_max_batch=mb, device_wiring=wiring)
    tag = "device search + %s wiring, b=%d" % ("device" if wiring else "host", mb)
    print("%s (%d host threads): %.2fs" % (tag, threads, time.time() - t0), flush=True)
    report(tag, ix); del ix
