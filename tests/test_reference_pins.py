"""The cereal-free parts of the reference that the product re-implements on the host, pinned against the EXECUTED reference
(oracle/_ref = the reference's own headers compiled where they lie, oracle/ref_cereal_free.cpp):

  * include/flatnav/util/Reordering.h  gOrder / rcmOrder  == reference Reordering.h:27-200 + GorderPriorityQueue.h:14-109 --
    the SAME PERMUTATION, element for element, ties in score / degree included ("pinned, not isomorphic");
  * include/flatnav/util/Multithreading.h  executeInParallel  == reference Multithreading.h:19-48 (every index exactly once,
    extra arguments forwarded, zero threads -> std::invalid_argument);
  * include/flatnav/util/Datatype.h  ordinals / names / sizes  == reference Datatype.h:11-186 (the ordinal is the first
    int32 of a saved index file and the C ABI's FNV_DTYPE_*).

The product side is compiled by this test from the product's headers (tests/cpp/host_utils_shim.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def own(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("shim") / "libhost_utils_shim.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "host_utils_shim.cpp"), "-o", out])
    L = C.CDLL(out)
    L.own_gorder.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
    L.own_rcm.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.own_execute_in_parallel.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    L.own_datatype_name.restype = C.c_char_p
    L.own_datatype_name.argtypes = [C.c_int]
    L.own_datatype_ordinal.argtypes = [C.c_char_p]
    for name in ("own_datatype_size", "own_datatype_ctype_bytes"):
        getattr(L, name).restype = C.c_uint64
        getattr(L, name).argtypes = [C.c_int]
    L.own_datatype_enum_bytes.restype = C.c_uint64
    return L


def _csr(table):
    offsets = np.zeros(len(table) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum([len(r) for r in table])
    flat = np.array([v for r in table for v in r] + [0], dtype=np.uint32)  # (+ one element: never an empty buffer)
    return flat, offsets


def _random_graph(rng, kind):
    """Out-degree tables as Index::getGraphOutdegreeTable leaves them (no self-loops in a row, ids < n), in several shapes:
    equal degrees everywhere (every comparison a tie), ragged rows, isolated nodes, a ring, duplicate edges."""
    n = int(rng.integers(2, 400))
    if kind == "regular":
        M = int(rng.integers(1, min(n, 9)))
        return [list(rng.choice([u for u in range(n) if u != v] or [v], size=min(M, n - 1), replace=False)) for v in range(n)]
    if kind == "ragged":
        return [list(rng.choice(n, size=int(rng.integers(0, min(n, 12))), replace=False)) for v in range(n)]
    if kind == "ring":
        return [[(v + 1) % n, (v + 2) % n] for v in range(n)]
    if kind == "sparse":  # many isolated nodes and components
        return [list(rng.choice(n, size=int(rng.integers(0, 2)), replace=False)) if rng.random() < 0.6 else [] for v in range(n)]
    return [list(rng.integers(0, n, size=int(rng.integers(0, 8)))) for v in range(n)]  # "dupes": repeated targets allowed


KINDS = ["regular", "ragged", "ring", "sparse", "dupes"]


def test_gorder_and_rcm_return_the_references_permutation(own, ref):
    if not hasattr(ref, "ref_gorder"):
        pytest.skip("oracle/_ref predates round 6 (rebuild it in the dev container: make -C oracle ref)")
    rng = np.random.default_rng(60)
    checked = 0
    for trial in range(60):
        table = _random_graph(rng, KINDS[trial % len(KINDS)])
        n = len(table)
        flat, offsets = _csr(table)
        for w in (5, 1, 0, 2, 17):  # Index.h:418,429 uses w = 5
            a, b = np.empty(n, np.uint32), np.empty(n, np.uint32)
            ref.ref_gorder(flat.ctypes.data, offsets.ctypes.data, n, w, a.ctypes.data)
            own.own_gorder(flat.ctypes.data, offsets.ctypes.data, n, w, b.ctypes.data)
            assert np.array_equal(a, b), (trial, KINDS[trial % len(KINDS)], n, w)
            assert sorted(a.tolist()) == list(range(n))
        a, b = np.empty(n, np.uint32), np.empty(n, np.uint32)
        ref.ref_rcm(flat.ctypes.data, offsets.ctypes.data, n, a.ctypes.data)
        own.own_rcm(flat.ctypes.data, offsets.ctypes.data, n, b.ctypes.data)
        assert np.array_equal(a, b), (trial, KINDS[trial % len(KINDS)], n)
        assert sorted(a.tolist()) == list(range(n))
        checked += 1
    assert checked == 60


def test_gorder_and_rcm_on_a_real_index_graph(own, ref, oracle_mod):
    # the out-degree table of a built index (Index.h:240-260: links that are not self-loops), 3000 nodes, M = 16
    if not hasattr(ref, "ref_gorder"):
        pytest.skip("oracle/_ref predates round 6")
    from flatnav_amd import datasets as ds

    X, _ = ds.sift_like(3000, 1)
    ix = oracle_mod.OracleIndex.create("l2", 128, 3000, 16)
    ix.add(X, 48)
    blob = np.asarray(ix.blob()).reshape(3000, ix.node_size)
    links = blob[:, ix.data_size:ix.data_size + 64].copy().view(np.uint32)
    table = [[int(v) for v in links[u] if v != u] for u in range(3000)]
    flat, offsets = _csr(table)
    for fn_ref, fn_own, args in ((ref.ref_gorder, own.own_gorder, (5,)), (ref.ref_rcm, own.own_rcm, ())):
        a, b = np.empty(3000, np.uint32), np.empty(3000, np.uint32)
        fn_ref(flat.ctypes.data, offsets.ctypes.data, 3000, *args, a.ctypes.data)
        fn_own(flat.ctypes.data, offsets.ctypes.data, 3000, *args, b.ctypes.data)
        assert np.array_equal(a, b)


def test_execute_in_parallel_covers_every_index_once(own, ref):
    if not hasattr(ref, "ref_execute_in_parallel"):
        pytest.skip("oracle/_ref predates round 6")
    for start, end, threads, extra in ((0, 1000, 4, 0), (7, 8, 3, 2), (5, 5, 2, 0), (9, 3, 2, 1), (0, 50000, 8, 3), (0, 13, 1, 0)):
        n = max(0, end - start)
        got = []
        for fn in (ref.ref_execute_in_parallel, own.own_execute_in_parallel):
            hits = np.zeros(max(1, n), dtype=np.uint32)
            assert fn(start, end, threads, extra, hits.ctypes.data) == 0
            got.append(hits)
        assert np.array_equal(got[0], got[1])
        assert (got[1][:n] == 1 + extra).all()  # each index handed out exactly once, the extra argument forwarded by value
    z = np.zeros(4, dtype=np.uint32)
    assert ref.ref_execute_in_parallel(0, 4, 0, 0, z.ctypes.data) == 1  # Multithreading.h:22-24
    assert own.own_execute_in_parallel(0, 4, 0, 0, z.ctypes.data) == 1
    assert not z.any()


def test_datatype_ordinals_names_and_sizes(own, ref):
    if not hasattr(ref, "ref_datatype_name"):
        pytest.skip("oracle/_ref predates round 6")
    from flatnav_amd import hip
    from oracle import oracle as orc

    assert ref.ref_datatype_enum_bytes() == own.own_datatype_enum_bytes() == 4  # the int32 at offset 0 of an index file
    for ordinal in range(0, 14):
        assert ref.ref_datatype_name(ordinal) == own.own_datatype_name(ordinal), ordinal
        assert ref.ref_datatype_size(ordinal) == own.own_datatype_size(ordinal), ordinal
        assert ref.ref_datatype_ctype_bytes(ordinal) == own.own_datatype_ctype_bytes(ordinal), ordinal
    for label in ("uint8", "uint16", "uint32", "uint64", "int8", "int16", "int32", "int64", "float16", "float32", "float64",
                  "undefined", "bfloat16", ""):
        assert ref.ref_datatype_ordinal(label.encode()) == own.own_datatype_ordinal(label.encode()), label
    # the three index element types: the ordinals the oracle, the ctypes binding and the C ABI use are the reference's
    for label, ordinal in (("uint8", 0), ("int8", 4), ("float32", 9)):
        assert ref.ref_datatype_ordinal(label.encode()) == ordinal == orc.DTYPE_ORD[label] == hip.DTYPE_ORD[label]


@pytest.mark.parametrize("methods", [["gorder"], ["rcm"], ["gorder", "rcm"]], ids=["gorder", "rcm", "gorder+rcm"])
def test_index_reorder_applies_the_references_permutation(ref, methods):
    # The whole product path Index::doGraphReordering -> relabel (reference Index.h:412-440, 872-926) through the Python
    # surface: after reorder(methods) node u sits at P[u] with every link rewritten through P and its label travelling along,
    # where P is what the REFERENCE's gOrder(table, 5) / rcmOrder(table) return on the table before.  No GPU involved.
    if not hasattr(ref, "ref_gorder"):
        pytest.skip("oracle/_ref predates round 6")
    from flatnav_amd import build_host

    build_host.build()
    import flatnav_amd as flatnav

    rng = np.random.default_rng(len(methods) * 7 + len(methods[0]))
    N, dim, M = 1500, 16, 8
    X = rng.integers(0, 12, (N, dim)).astype(np.uint8)  # tie-heavy: many equal degrees and equal scores
    ix = flatnav.index.create("l2", dim, N, M, flatnav.data_type.DataType.uint8)
    ix.add(X, 32, labels=[int(v) for v in rng.permutation(N) + 100])
    node = ix._node_size_bytes

    def rows():
        blob = np.asarray(ix._raw_blob()).reshape(N, node)
        return (blob[:, :dim].copy(), blob[:, dim:dim + 4 * M].copy().view(np.uint32), blob[:, dim + 4 * M:].copy().view(np.int32).ravel())

    vec, links, labels = rows()
    for method in methods:
        table = ix.get_graph_outdegree_table()
        assert table == [[int(v) for v in links[u] if v != u] for u in range(N)]  # Index.h:240-260
        flat, offsets = _csr(table)
        P = np.empty(N, np.uint32)
        if method == "gorder":
            ref.ref_gorder(flat.ctypes.data, offsets.ctypes.data, N, 5, P.ctypes.data)
        else:
            ref.ref_rcm(flat.ctypes.data, offsets.ctypes.data, N, P.ctypes.data)
        want_vec, want_links, want_labels = np.empty_like(vec), np.empty_like(links), np.empty_like(labels)
        want_vec[P], want_links[P], want_labels[P] = vec, P[links], labels
        ix.reorder([method])
        vec, links, labels = rows()
        assert np.array_equal(vec, want_vec) and np.array_equal(labels, want_labels)
        assert np.array_equal(links, want_links)
