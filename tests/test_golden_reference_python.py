"""flatnav_amd.io and the benchmark harness' metric definitions against outputs of the REFERENCE'S OWN Python code
(experiments/data_loader.py, experiments/plotting/metrics.py), captured in the dev container by
tests/golden/make_reference_python_golden.py and committed as data under tests/golden/refpy/."""
import os

import numpy as np
import pytest

from flatnav_amd import datasets as ds, io as fio

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "refpy")


@pytest.fixture(scope="module")
def exp():
    return np.load(os.path.join(G, "expected.npz"))


def test_vecs_files_like_the_reference_readers(exp):
    # reference ranges are 1-based inclusive (data_loader.py:7-46); rows=(a, b) here is 0-based half-open
    assert np.array_equal(fio.read_vecs(os.path.join(G, "base.bvecs")), exp["bvecs_all"])
    assert np.array_equal(fio.read_vecs(os.path.join(G, "base.bvecs"), rows=(2, 10)), exp["bvecs_3_10"])
    assert np.array_equal(fio.read_vecs(os.path.join(G, "gt.ivecs")), exp["ivecs_all"])
    assert np.array_equal(fio.read_vecs(os.path.join(G, "gt.ivecs"), rows=(4, 40)), exp["ivecs_5_40"])


@pytest.mark.parametrize("ext", ["fbin", "u8bin", "i8bin"])
def test_bin_datasets_like_the_reference_loader(exp, ext):
    for tag, rows in (("all", None), ("4_11", (4, 11))):
        X, Q, GT = fio.load_dataset(os.path.join(G, "base." + ext), os.path.join(G, "query." + ext),
                                    os.path.join(G, "gt.bin"), rows=rows)
        assert X.dtype == exp["%s_train_%s" % (ext, tag)].dtype
        assert np.array_equal(X, exp["%s_train_%s" % (ext, tag)])
        assert np.array_equal(Q, exp["%s_queries_%s" % (ext, tag)])
        assert np.array_equal(GT, exp["%s_gt_%s" % (ext, tag)].astype(np.int32))
    ids, dist = fio.read_ground_truth_bin(os.path.join(G, "gt.bin"))
    assert np.array_equal(ids, exp["gtbin_ids"]) and np.array_equal(dist, exp["gtbin_dist"])
    assert list(ids.shape) == exp["gtbin_shape"].tolist()


def test_npy_dataset_like_the_reference_loader(exp):
    for tag, rows in (("all", None), ("2_9", (2, 9))):
        X, Q, GT = fio.load_dataset(os.path.join(G, "train.npy"), os.path.join(G, "test.npy"),
                                    os.path.join(G, "neighbors.npy"), rows=rows)
        # the reference casts to float32 / int32 (data_loader.py:92-107); the index does the same cast on add/search
        assert np.array_equal(np.asarray(X, dtype=np.float32), exp["npy_train_" + tag])
        assert np.array_equal(np.asarray(Q, dtype=np.float32), exp["npy_queries_" + tag])
        assert GT.dtype == np.int32 and np.array_equal(GT, exp["npy_gt_" + tag])


def test_harness_metrics_like_the_reference_definitions(exp):
    # recall (plotting/metrics.py:53-66): hits of each returned id in the truth SET, / k, mean over queries -- a
    # duplicate id in a result row counts twice there; recall_at_k intersects sets, so compare on duplicate-free rows
    found, truth = exp["recall_found"], exp["recall_truth"]
    ref_recall = float(exp["recall_value"])
    per_row = np.array([sum(1 for x in f if x in set(t.tolist())) / 10 for f, t in zip(found, truth)])
    assert abs(per_row.mean() - ref_recall) < 1e-12
    nodup = np.array([len(set(f.tolist())) == 10 for f in found])
    assert nodup.sum() >= 30
    ours = np.array([ds.recall_at_k(f[None, :], t[None, :]) for f, t in zip(found[nodup], truth[nodup])])
    assert np.allclose(ours, per_row[nodup])
    lat = exp["latencies"]
    for name, q in (("latency_p50", 50), ("latency_p90", 90), ("latency_p95", 95), ("latency_p99", 99), ("latency_p999", 99.9)):
        assert float(exp[name]) == float(np.percentile(lat, q) * 1e3)  # tools/run_benchmark.py computes exactly this
    assert float(exp["qps"]) == len(lat) / float(lat.sum())
    assert float(exp["distance_computations"]) == 123456 / 997
