"""Host API (header-only C++17 flatnav::Index through the pybind11 module) without a GPU:
construction parity with the oracle, file-format interchange, API surface and error behaviour of the
reference's Python binding (python-bindings/src/flatnav/bindings.cpp:426-539; unit_tests/test_index.py)."""
import os
import struct

import numpy as np
import pytest

from flatnav_amd import datasets as ds


@pytest.fixture(scope="module")
def flatnav():
    from flatnav_amd import build_host

    build_host.build()
    import flatnav_amd

    return flatnav_amd


CASES = [("l2", "float32"), ("angular", "float32"), ("l2", "uint8"), ("angular", "uint8"), ("l2", "int8"),
         ("angular", "int8")]


def _data(dt, n, dim, seed, small=False):
    rng = np.random.default_rng(seed)
    if dt == "float32":
        return rng.integers(0, 16 if small else 256, (n, dim)).astype(np.float32)
    if dt == "uint8":
        return rng.integers(0, 16 if small else 256, (n, dim)).astype(np.uint8)
    return rng.integers(-8 if small else -128, 8 if small else 128, (n, dim)).astype(np.int8)


def test_module_surface(flatnav):
    # names of the reference package (python-bindings/src/flatnav/__init__.py:1-35)
    assert {"IndexL2Float", "IndexIPFloat", "IndexL2Uint8", "IndexIPUint8", "IndexL2Int8", "IndexIPInt8",
            "create"} <= set(dir(flatnav.index))
    DT = flatnav.data_type.DataType
    assert {DT.float32.name, DT.int8.name, DT.uint8.name} == {"float32", "int8", "uint8"}
    assert int(DT.uint8) == 0 and int(DT.int8) == 4 and int(DT.float32) == 9  # file-format ordinals
    assert flatnav.MetricType.L2 != flatnav.MetricType.IP
    import sys

    assert sys.modules["flatnav_amd.index"] is flatnav.index
    assert isinstance(flatnav.__version__, str)


@pytest.mark.parametrize("metric,dt", CASES)
def test_single_thread_build_equals_oracle_graph(flatnav, oracle_mod, tmp_path, metric, dt):
    N, dim, M = 1200, 40, 16
    X = _data(dt, N, dim, 11, small=(metric == "angular"))
    DT = getattr(flatnav.data_type.DataType, dt)
    ix = flatnav.index.create(distance_type=metric, index_data_type=DT, dim=dim, dataset_size=N, max_edges_per_node=M)
    assert type(ix).__name__ == "Index%s%s" % ("L2" if metric == "l2" else "IP",
                                                {"float32": "Float", "uint8": "Uint8", "int8": "Int8"}[dt])
    assert ix.max_edges_per_node == M and ix.num_threads == 1
    ix.add(data=X, ef_construction=64)
    o = oracle_mod.OracleIndex.create(metric, dim, N, M, dt)
    o.add(X, 64)
    assert np.array_equal(np.asarray(ix._raw_blob()), o.blob())  # identical node store (vectors, links, labels)
    # file format: byte-identical files, loadable in both directions (SURVEY.md App. B layout)
    p1, p2 = str(tmp_path / "product.bin"), str(tmp_path / "oracle.bin")
    ix.save(p1)
    o.save(p2)
    assert open(p1, "rb").read() == open(p2, "rb").read()
    hdr = struct.unpack("<i7Q", open(p1, "rb").read(60))
    assert hdr == (int(DT), M, X.dtype.itemsize * dim, X.dtype.itemsize * dim + 4 * M + 4, N, N, dim,
                   X.dtype.itemsize * dim)
    loaded = type(ix).load_index(p2)
    assert loaded.max_edges_per_node == M
    assert loaded.num_threads == max(1, (os.cpu_count() or 2) // 2)  # Index.h:467 of the reference
    assert np.array_equal(np.asarray(loaded._raw_blob()), o.blob())
    o2 = oracle_mod.OracleIndex.load(p1, metric)
    assert np.array_equal(o2.blob(), o.blob())


def test_single_thread_build_equals_oracle_graph_on_random_shapes(flatnav, oracle_mod, tmp_path):
    # The host builder against the oracle's graph bytes over randomly drawn shapes: element type, metric, row width, M,
    # ef_construction (also smaller than M/2: the beam is then wired as its heap pops it), tie density.
    rng = np.random.default_rng(int(os.environ.get("FNV_FUZZ_SEED", "404")))
    for trial in range(int(os.environ.get("FNV_FUZZ_TRIALS", "24"))):
        dt = ["float32", "uint8", "int8"][trial % 3]
        metric = ["l2", "angular"][int(rng.integers(0, 2))]
        dim = int(rng.choice([3, 16, 33, 100]))
        M = int(rng.choice([2, 4, 8, 16, 32]))
        N = int(rng.integers(50, 900))
        efc = int(rng.choice([3, 10, 40, 100]))
        hi = int(rng.choice([2, 4, 16, 100]))
        lo = -hi // 2 if dt == "int8" else 0
        X = rng.integers(lo, lo + hi, (N, dim)).astype(dt)
        kw = {} if dt == "float32" else {"index_data_type": getattr(flatnav.data_type.DataType, dt)}
        ix = flatnav.index.create(metric, dim, N, M, **kw)
        ix.set_num_threads(1)
        ix.add(X, efc)
        o = oracle_mod.OracleIndex.create(metric, dim, N, M, dt)
        o.add(X, efc)
        what = "trial %d: %s %s d=%d M=%d N=%d efc=%d hi=%d" % (trial, dt, metric, dim, M, N, efc, hi)
        assert np.array_equal(np.asarray(ix._raw_blob()), o.blob()), what
        if trial % 3 == 0:  # the files: byte-identical, and each side loads the other's
            p1, p2 = str(tmp_path / "a.bin"), str(tmp_path / "b.bin")
            ix.save(p1)
            o.save(p2)
            assert open(p1, "rb").read() == open(p2, "rb").read(), what
            assert np.array_equal(np.asarray(type(ix).load_index(p2)._raw_blob()), o.blob()), what
            assert np.array_equal(oracle_mod.OracleIndex.load(p1, "l2" if metric == "l2" else "ip").blob(), o.blob()), what


def test_forcecast_and_labels(flatnav, oracle_mod):
    X = _data("float32", 500, 16, 3)
    ix = flatnav.index.create("l2", 16, 500, 8)
    labels = list(range(1000, 1500))
    ix.add(X.astype(np.float64), 32, labels=labels)  # float64 input is force-cast (bindings.cpp:36-51)
    o = oracle_mod.OracleIndex.create("l2", 16, 500, 8)
    o.add(X, 32, labels=np.array(labels, dtype=np.int32))
    assert np.array_equal(np.asarray(ix._raw_blob()), o.blob())
    table = ix.get_graph_outdegree_table()
    assert len(table) == 500 and all(len(r) <= 8 for r in table) and all(i not in r for i, r in enumerate(table))


def test_multithreaded_build_is_a_valid_graph(flatnav, oracle_mod):
    X, Q = ds.sift_like(6000, 200)
    ix = flatnav.index.create("l2", 128, 6000, 16)
    ix.set_num_threads(min(8, os.cpu_count() or 1))
    ix.add(X, 64)
    o = oracle_mod.OracleIndex.from_blob("l2", "float32", 128, 6000, 6000, 16, np.asarray(ix._raw_blob()))
    _, l = o.search(Q, 10, 100)
    assert ds.recall_at_k(l, ds.exact_topk_l2(X, Q, 10)) > 0.95


def test_errors_match_reference(flatnav):
    with pytest.raises(ValueError):  # bindings.cpp:397-407
        flatnav.index.create("cosine", 8, 10, 4)
    ix = flatnav.index.create("l2", 8, 10, 4)
    X = _data("float32", 11, 8, 0)
    with pytest.raises(ValueError):
        ix.add(X[:, :5], 10)  # wrong dimension (bindings.cpp:74-84)
    with pytest.raises(ValueError):
        ix.add(X[:4], 10, num_initializations=0)  # Index.h:303-305
    with pytest.raises(ValueError):
        ix.add(X[:4], 10, labels=[1, 2])  # "Incorrect number of labels."
    with pytest.raises(RuntimeError):
        ix.add(X, 10)  # 11 > dataset_size (Index.h:355-360)
    with pytest.raises(ValueError):
        ix.search(X[:2], 3, 10, num_initializations=0)  # Index.h:847-849, raised before touching the device
    with pytest.raises(ValueError):
        ix.search(X[:2, :3], 3, 10)
    with pytest.raises(ValueError):
        ix.search_single(X[:2], 3, 10)  # needs a 1-D query
    with pytest.raises(ValueError):
        ix.set_num_threads(0)  # Index.h:493-497
    with pytest.raises(ValueError):
        ix.reorder(["metis"])  # bindings.cpp:290-292
    with pytest.raises(RuntimeError):
        type(ix).load_index("/nonexistent/index.bin")  # Index.h:445-447
    with pytest.raises(RuntimeError):
        ix.build_graph_links("/nonexistent/graph.mtx")  # Index.h:189-191


def test_build_graph_links_and_reorder(flatnav, tmp_path):
    # allocate_nodes + build_graph_links (import of an HNSW base layer, Index.h:187-238; the size line is
    # "<nodes> <nodes> <M>", 1-based "u v" pairs fill the first free slot of u)
    N, M = 50, 4
    X = _data("float32", N, 8, 5)
    ix = flatnav.index.create("l2", 8, N, M)
    assert ix.allocate_nodes(X) is ix
    mtx = tmp_path / "g.mtx"
    edges = [(u, (u + k) % N) for u in range(N) for k in (1, 2, 3)]
    mtx.write_text("%%MatrixMarket matrix coordinate pattern general\n%d %d %d\n" % (N, N, M) +
                   "".join("%d %d\n" % (u + 1, v + 1) for u, v in edges))
    ix.build_graph_links(str(mtx))
    table = ix.get_graph_outdegree_table()
    assert all(table[u] == [(u + 1) % N, (u + 2) % N, (u + 3) % N] for u in range(N))
    bad = tmp_path / "bad.mtx"
    bad.write_text("%d %d %d\n" % (N, N, M + 1))
    with pytest.raises(RuntimeError):
        ix.build_graph_links(str(bad))
    # reorder keeps the graph isomorphic: same multiset of out-degrees, labels travel with the nodes
    before = sorted(len(r) for r in table)
    ix.reorder(["gorder", "rcm"])
    after = ix.get_graph_outdegree_table()
    assert sorted(len(r) for r in after) == before
    blob = np.asarray(ix._raw_blob()).reshape(N, ix._node_size_bytes)
    labels = blob[:, -4:].copy().view(np.int32).ravel()
    assert sorted(labels.tolist()) == list(range(N))
