#!/usr/bin/env python3
"""Benchmark harness with the reference's metric definitions (experiments/run-benchmark.py:38-124,
experiments/plotting/metrics.py:53-132), own implementation, for the GPU index.

Two regimes per ef_search:
  * batched  -- index.search(all queries): how the GPU is meant to be used (qps_batched);
  * per query -- index.search_single in a Python loop, the reference's protocol: qps = n / sum(latencies),
    latency_p50/p90/p95/p99/p999 (ms), distance_computations per query (collect_stats=True).
Writes a metrics.json keyed like the reference's (dataset_name -> list of experiment dicts).

Data: --train/--queries/--gtruth files (.npy, *vecs, *bin via flatnav_amd.io) or --synthetic sift|lowrank768.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import flatnav_amd as flatnav  # noqa: E402
from flatnav_amd import datasets as ds  # noqa: E402
from flatnav_amd import io as fio  # noqa: E402


def recall(found, truth, k):
    hits = sum(len(set(f[:k].tolist()) & set(t[:k].tolist())) for f, t in zip(found, truth))
    return hits / (k * len(truth))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train"), ap.add_argument("--queries"), ap.add_argument("--gtruth")
    ap.add_argument("--synthetic", choices=["sift", "lowrank768"], default=None)
    ap.add_argument("--n", type=int, default=200_000)
    ap.add_argument("--num-queries", type=int, default=2000)
    ap.add_argument("--metric", default="l2", choices=["l2", "angular"])
    ap.add_argument("--dataset-name", default="synthetic")
    ap.add_argument("--num-node-links", type=int, nargs="+", default=[32])
    ap.add_argument("--ef-construction", type=int, nargs="+", default=[100])
    ap.add_argument("--ef-search", type=int, nargs="+", default=[100, 200, 300])
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--num-build-threads", type=int, default=0)
    ap.add_argument("--single-query-samples", type=int, default=500)
    ap.add_argument("--metrics-file", default="metrics.json")
    a = ap.parse_args()

    if a.synthetic:
        X, Q = (ds.sift_like(a.n, a.num_queries) if a.synthetic == "sift"
                else ds.lowrank_normalized(a.n, a.num_queries))
        metric = "l2" if a.synthetic == "sift" else "angular"
        G = (ds.exact_topk_l2 if metric == "l2" else ds.exact_topk_ip)(X, Q, a.k)
    else:
        X, Q, G = fio.load_dataset(a.train, a.queries, a.gtruth, normalize=False)
        metric = a.metric
        Q, G = Q[:a.num_queries], G[:a.num_queries]
    X = np.ascontiguousarray(X, dtype=np.float32)
    Q = np.ascontiguousarray(Q, dtype=np.float32)
    threads = a.num_build_threads or ds.effective_cpus()
    experiments = []
    for M in a.num_node_links:
        for efc in a.ef_construction:
            index = flatnav.index.create(distance_type=metric, index_data_type=flatnav.data_type.DataType.float32,
                                         dim=X.shape[1], dataset_size=len(X), max_edges_per_node=M, collect_stats=True)
            index.set_num_threads(threads)
            t0 = time.time()
            index.add(data=X, ef_construction=efc)
            build_time = time.time() - t0
            index.get_query_distance_computations()  # reset (the counter includes the build)
            for ef in a.ef_search:
                index.search(Q[:64], a.k, ef)  # warm-up (also uploads the index)
                index.get_query_distance_computations()
                t0 = time.perf_counter()
                _, labels = index.search(queries=Q, K=a.k, ef_search=ef, num_initializations=100)
                batched = len(Q) / (time.perf_counter() - t0)
                dist_comps = index.get_query_distance_computations() / len(Q)
                lat = []
                for q in Q[:a.single_query_samples]:
                    t0 = time.perf_counter()
                    index.search_single(query=q, K=a.k, ef_search=ef, num_initializations=100)
                    lat.append(time.perf_counter() - t0)
                lat = np.array(lat)
                m = {
                    "recall": recall(labels, G, a.k),
                    "qps_batched": batched,
                    "qps": len(lat) / lat.sum(),  # reference definition: n / sum of per-query latencies
                    "latency_p50": float(np.percentile(lat, 50) * 1e3),
                    "latency_p90": float(np.percentile(lat, 90) * 1e3),
                    "latency_p95": float(np.percentile(lat, 95) * 1e3),
                    "latency_p99": float(np.percentile(lat, 99) * 1e3),
                    "latency_p999": float(np.percentile(lat, 99.9) * 1e3),
                    "distance_computations": dist_comps,
                    "build_time": build_time,
                    "index_size": len(X) * index._node_size_bytes,
                    "node_links": M, "ef_construction": efc, "ef_search": ef, "k": a.k,
                }
                experiments.append(m)
                print(json.dumps(m), flush=True)
    out = {a.dataset_name: experiments}
    with open(a.metrics_file, "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
