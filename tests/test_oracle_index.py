"""Oracle self-consistency in the shape of the reference's own integration test
(include/flatnav/tests/test_serialization.cpp:36-176) plus API error behaviour."""
import os
import struct

import numpy as np
import pytest

from flatnav_amd import datasets as ds

CASES = [("l2", "float32"), ("ip", "float32"), ("l2", "uint8"), ("ip", "uint8"), ("l2", "int8"), ("ip", "int8")]


def _data(dt, n, dim, seed):
    rng = np.random.default_rng(seed)
    if dt == "float32":
        return rng.random((n, dim), dtype=np.float32)
    if dt == "uint8":
        return rng.integers(0, 256, (n, dim)).astype(np.uint8)
    return rng.integers(-128, 128, (n, dim)).astype(np.int8)


@pytest.mark.parametrize("metric,dt", CASES)
def test_save_load_identical_search(oracle_mod, tmp_path, metric, dt):
    N, dim, M = 1500, 48, 16
    X = _data(dt, N, dim, 1)
    Q = _data(dt, 100, dim, 2)
    ix = oracle_mod.OracleIndex.create(metric, dim, N, M, dt)
    ix.add(X, 100)
    path = str(tmp_path / "index.bin")
    ix.save(path)
    # Appendix B of SURVEY.md: 60-byte header + node_size * max_nodes, node = data + 4M + 4
    assert ix.node_size == ix.data_size + 4 * M + 4
    assert os.path.getsize(path) == 60 + ix.node_size * N
    with open(path, "rb") as f:
        hdr = struct.unpack("<i7Q", f.read(60))
    assert hdr == (oracle_mod.DTYPE_ORD[dt], M, ix.data_size, ix.node_size, N, N, dim, ix.data_size)
    ix2 = oracle_mod.OracleIndex.load(path, metric)
    assert (ix2.M, ix2.dim, ix2.dtype, ix2.cur_nodes) == (M, dim, dt, N)
    d1, l1 = ix.search(Q, 10, 50)
    d2, l2 = ix2.search(Q, 10, 50)
    assert np.array_equal(d1, d2) and np.array_equal(l1, l2)


def test_search_on_reference_distance_kernels_is_identical_on_integer_data(oracle_mod, ref):
    X, Q = ds.sift_like(4000, 200)
    ix = oracle_mod.OracleIndex.create("l2", 128, 4000, 32)
    ix.add(X, 100)
    blob_own = ix.blob().copy()
    d1, l1, s1 = ix.search(Q, 10, 100, stats=True)
    # same graph, reference's compiled AVX-512 kernel as the distance
    assert ix.use_reference_distance(True)
    d2, l2, s2 = ix.search(Q, 10, 100, stats=True)
    assert np.array_equal(d1, d2) and np.array_equal(l1, l2)
    assert np.array_equal(s1["n_dist"], s2["n_dist"]) and np.array_equal(s1["n_hops"], s2["n_hops"])
    # and the single-thread BUILD is identical too (construction goes through the same distances)
    ix3 = oracle_mod.OracleIndex.create("l2", 128, 4000, 32)
    ix3.use_reference_distance(True)
    ix3.add(X, 100)
    assert np.array_equal(ix3.blob(), blob_own)


def test_recall_and_counters(oracle_mod):
    X, Q = ds.sift_like(5000, 100)
    ix = oracle_mod.OracleIndex.create("l2", 128, 5000, 32)
    ix.add(X, 100)
    d, l, st = ix.search(Q, 10, 100, stats=True)
    gt = ds.exact_topk_l2(X, Q, 10)
    assert ds.recall_at_k(l, gt) > 0.97
    assert (np.diff(d, axis=1) >= 0).all()
    assert (st["count"] == 10).all()
    # hops ~ ef, one candidate popped per hop, admissions bound the candidate heap
    assert (st["n_hops"] >= 1).all() and (st["max_cand"] <= st["n_admit"] + 1).all()


def test_entry_point_scan_counts(oracle_mod):
    # Index.h:851-861: step = max(1, N / n_init); scans ceil(N/step) nodes; K > N returns < K results
    X = _data("float32", 150, 8, 3)
    ix = oracle_mod.OracleIndex.create("l2", 8, 200, 8)
    ix.add(X, 50)
    d, l, st = ix.search(X[:5], 200, 300, stats=True)
    assert (st["count"] <= 150).all()
    assert (l[:, 0] == np.arange(5)).all() and (d[:, 0] == 0).all()


def test_errors(oracle_mod):
    ix = oracle_mod.OracleIndex.create("l2", 8, 10, 4)
    X = _data("float32", 11, 8, 0)
    with pytest.raises(ValueError):
        ix.add(X[:5], 10, num_initializations=0)
    with pytest.raises(RuntimeError):
        ix.add(X, 10)  # 11 > max_nodes
    with pytest.raises(ValueError):
        ix.search(X[:2], 3, 10, num_initializations=0)
    with pytest.raises(ValueError):
        ix.search(X[:2, :4], 3, 10)
    with pytest.raises(RuntimeError):
        oracle_mod.OracleIndex.load("/nonexistent/file.bin", "l2")


def test_multithreaded_search_equals_single(oracle_mod):
    X, Q = ds.sift_like(3000, 300)
    ix = oracle_mod.OracleIndex.create("l2", 128, 3000, 16)
    ix.add(X, 64)
    a = ix.search(Q, 10, 64, threads=1)
    b = ix.search(Q, 10, 64, threads=4)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
