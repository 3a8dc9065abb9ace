#!/usr/bin/env python3
"""Generates tests/golden/refpy/: small dataset files in every format the reference's harness reads, and what the
REFERENCE'S OWN Python code returns for them -- executed here, in the dev container, by importing
/root/reference/experiments/data_loader.py and /root/reference/experiments/plotting/metrics.py (no reference source is
copied; the fixtures are data: input files + expected arrays).  tests/test_golden_reference_python.py then pins
flatnav_amd.io and the harness' metric definitions against these outputs wherever the tests run.

    python tests/golden/make_reference_python_golden.py      (needs /root/reference)
"""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "refpy")
REF = "/root/reference/experiments"


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def write_vecs(path, a):
    n, d = a.shape
    rec = np.empty((n, 4 + d * a.dtype.itemsize), dtype=np.uint8)
    rec[:, :4] = np.array([d], dtype=np.int32).view(np.uint8)
    rec[:, 4:] = np.ascontiguousarray(a).view(np.uint8).reshape(n, -1)
    rec.tofile(path)


def write_bin(path, a):
    with open(path, "wb") as f:
        np.array(a.shape, dtype=np.uint32).tofile(f)
        np.ascontiguousarray(a).tofile(f)


def main():
    os.makedirs(OUT, exist_ok=True)
    dl = load("ref_data_loader", os.path.join(REF, "data_loader.py"))
    metrics = load("ref_metrics", os.path.join(REF, "plotting", "metrics.py"))
    rng = np.random.default_rng(20260)
    exp = {}
    p = lambda name: os.path.join(OUT, name)

    # TEXMEX *vecs (reference readers take 1-based inclusive ranges)
    bv = rng.integers(0, 256, (23, 12)).astype(np.uint8)
    iv = rng.integers(0, 1000, (23, 9)).astype(np.int32)
    write_vecs(p("base.bvecs"), bv)
    write_vecs(p("gt.ivecs"), iv)
    exp["bvecs_all"] = dl.read_bvecs_file(p("base.bvecs"))
    exp["bvecs_3_10"] = dl.read_bvecs_file(p("base.bvecs"), (3, 10))
    exp["ivecs_all"] = dl.read_ivecs_file(p("gt.ivecs"))
    exp["ivecs_5_40"] = dl.read_ivecs_file(p("gt.ivecs"), (5, 40))  # end clamps to the file

    # big-ann-benchmarks *bin + ground truth
    ids = rng.integers(0, 17, (5, 4)).astype(np.uint32)
    dist = rng.random((5, 4), dtype=np.float32)
    with open(p("gt.bin"), "wb") as f:
        np.array(ids.shape, dtype=np.uint32).tofile(f)
        ids.tofile(f)
        dist.tofile(f)
    for ext, dt in ((".fbin", np.float32), (".u8bin", np.uint8), (".i8bin", np.int8)):
        base = (rng.random((17, 6)) * 200 - 100).astype(dt) if dt != np.uint8 else rng.integers(0, 256, (17, 6)).astype(dt)
        q = (rng.random((5, 6)) * 200 - 100).astype(dt) if dt != np.uint8 else rng.integers(0, 256, (5, 6)).astype(dt)
        write_bin(p("base" + ext), base)
        write_bin(p("query" + ext), q)
        for tag, rg in (("all", None), ("4_11", (4, 11))):
            loader = dl.get_data_loader(train_dataset_path=p("base" + ext), queries_path=p("query" + ext),
                                        ground_truth_path=p("gt.bin"), range=rg)
            t, qq, g = loader.load_data()
            exp["%s_train_%s" % (ext[1:], tag)] = np.array(t)
            exp["%s_queries_%s" % (ext[1:], tag)] = np.array(qq)
            exp["%s_gt_%s" % (ext[1:], tag)] = np.array(g)
    gi, gd, nq, k = dl.BinaryDatasetLoader(dtype=np.float32, train_dataset_path=p("base.fbin"), queries_path=p("query.fbin"),
                                           ground_truth_path=p("gt.bin")).load_ground_truth(p("gt.bin"))
    exp["gtbin_ids"], exp["gtbin_dist"], exp["gtbin_shape"] = np.array(gi), np.array(gd), np.array([nq, k])

    # .npy (the loader casts float64 -> float32, int64 -> int32)
    np.save(p("train.npy"), rng.random((11, 5)))
    np.save(p("test.npy"), rng.random((3, 5)))
    np.save(p("neighbors.npy"), rng.integers(0, 11, (3, 4)))
    for tag, rg in (("all", None), ("2_9", (2, 9))):
        t, qq, g = dl.get_data_loader(train_dataset_path=p("train.npy"), queries_path=p("test.npy"),
                                      ground_truth_path=p("neighbors.npy"), range=rg).load_data()
        exp["npy_train_" + tag], exp["npy_queries_" + tag], exp["npy_gt_" + tag] = np.array(t), np.array(qq), np.array(g)

    # harness metrics (plotting/metrics.py): recall with and without duplicates / misses, percentiles, ratios
    truth = np.stack([rng.permutation(50)[:10] for _ in range(40)])  # distinct ids per row
    found = truth.copy()
    found[::3, 2:6] = rng.integers(50, 99, (14, 4))   # misses
    found[1::5, 7] = found[1::5, 6]                    # a duplicate id inside a result row
    exp["recall_truth"], exp["recall_found"] = truth, found
    mm = metrics.metric_manager
    exp["recall_value"] = np.array(mm.compute_metric("recall", queries=found, ground_truth=truth, top_k_indices=found, k=10))
    lat = rng.random(997) * 1e-3
    exp["latencies"] = lat
    for name in ("latency_p50", "latency_p90", "latency_p95", "latency_p99", "latency_p999"):
        exp[name] = np.array(mm.compute_metric(name, latencies=lat))
    exp["qps"] = np.array(mm.compute_metric("qps", querying_time=float(lat.sum()), num_queries=len(lat)))
    exp["distance_computations"] = np.array(mm.compute_metric("distance_computations", distance_computations=123456, num_queries=997))
    np.savez(p("expected.npz"), **exp)
    print("wrote", OUT, sorted(os.listdir(OUT)))


if __name__ == "__main__":
    sys.exit(main())
