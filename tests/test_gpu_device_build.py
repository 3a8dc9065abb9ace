"""Device-assisted construction: index.add(..., device=True) runs the insertions' beam searches
on the GPU in batches (reference insertion rule: Index.h:353-378 + selectNeighbors/connectNeighbors :714-834).
The graph is of the same family, not byte-identical (neither is the reference's own multi-threaded build), so
the checks are: structural validity, search quality equal to the host builder's, and -- the part that must be
exact -- the incrementally maintained device copy answering exactly like the CPU oracle on the host blob."""
import numpy as np
import pytest

from flatnav_amd import datasets as ds, hip

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def flatnav():
    import flatnav_amd

    return flatnav_amd


def _links(index, n, node_size, data_size, M):
    blob = np.asarray(index._raw_blob())[: n * node_size].reshape(n, node_size)
    return blob[:, data_size:data_size + 4 * M].copy().view(np.uint32)


def _check_graph(index, n, M):
    L = _links(index, n, index._node_size_bytes, index._data_size_bytes, M)
    assert (L < n).all()
    own = np.arange(n, dtype=np.uint32)[:, None]
    real = L != own
    assert real[1:].any(axis=1).all()  # every node but possibly the first is wired to something
    srt = np.sort(np.where(real, L, np.uint32(0xFFFFFFFF)), axis=1)
    dup = (srt[:, 1:] == srt[:, :-1]) & (srt[:, 1:] != 0xFFFFFFFF)
    assert not dup.any()  # no repeated neighbour in a row
    return real.sum(axis=1).mean()


@pytest.mark.parametrize("wiring", [True, False], ids=["device_wiring", "host_wiring"])
@pytest.mark.parametrize("dt", ["float32", "uint8"])
def test_device_build_matches_host_build_quality(flatnav, oracle_mod, dt, wiring):
    N, NQ, M, K = 20000, 500, 32, 10
    X, Q = ds.sift_like(N, NQ)
    X, Q = X.astype(dt), Q.astype(dt)
    gt = ds.exact_topk_l2(X.astype(np.float32), Q.astype(np.float32), K)
    DT = getattr(flatnav.data_type.DataType, dt)
    host = flatnav.index.create("l2", 128, N, M, DT)
    host.set_num_threads(4)
    host.add(X, 100)
    dev = flatnav.index.create("l2", 128, N, M, DT, collect_stats=True)
    dev.set_num_threads(4)
    dev.add(X, 100, device=True, device_max_batch=2048, device_wiring=wiring)
    assert dev._cur_num_nodes == N
    deg = _check_graph(dev, N, M)
    assert deg > 0.9 * _check_graph(host, N, M)
    assert dev.get_query_distance_computations() > N * 100  # insertions are counted like the reference counts them
    for ef in (50, 100):
        _, lh = host.search(Q, K, ef)
        dd, ld = dev.search(Q, K, ef)  # answered by the device copy that was maintained batch by batch
        assert ds.recall_at_k(ld, gt) > ds.recall_at_k(lh, gt) - 0.01
        o = oracle_mod.OracleIndex.from_blob("l2", dt, 128, N, N, M, np.asarray(dev._raw_blob()))
        od, ol = o.search(Q, K, ef)
        assert (ol == ld).all() and (od == dd).all()


def test_device_build_appends_to_an_existing_graph(flatnav, oracle_mod):
    N, NQ, M, K = 12000, 1000, 16, 10
    X, Q = ds.lowrank_normalized(N, NQ, 100, 24, 5)
    ix = flatnav.index.create("angular", 100, N, M, flatnav.data_type.DataType.float32)
    ix.set_num_threads(4)
    ix.add(X[:5000], 64, labels=list(range(100000, 105000)))
    ix.search(Q, K, 64)  # uploads the 5000-node graph: the builder must replace that device copy
    ix.add(X[5000:], 64, labels=list(range(200000, 200000 + N - 5000)), device=True)
    assert ix._cur_num_nodes == N
    _check_graph(ix, N, M)
    d, l = ix.search(Q, K, 100)
    gt = ds.exact_topk_ip(X, Q, K)
    lab = np.concatenate([np.arange(100000, 105000), np.arange(200000, 200000 + N - 5000)])
    assert ds.recall_at_k(l, lab[gt]) > 0.9
    o = oracle_mod.OracleIndex.from_blob("angular", "float32", 100, N, N, M, np.asarray(ix._raw_blob()))
    od, ol = o.search(Q, K, 100)
    assert (ol == l).all(axis=1).mean() >= 0.999  # float: rare rounding flips allowed, as in the other parity tests
    # a later host insertion invalidates the device copy again and still works
    with pytest.raises(RuntimeError):
        ix.add(X[:1], 64, device=True)  # full
    with pytest.raises(ValueError):
        flatnav.index.create("l2", 100, 10, M).add(X[:5], 64, 0, device=True)


def test_device_build_int8_inner_product(flatnav, oracle_mod):
    rng = np.random.default_rng(3)
    N, NQ, M, K = 6000, 200, 16, 10
    X = rng.integers(-20, 21, (N, 48)).astype(np.int8)
    Q = rng.integers(-20, 21, (NQ, 48)).astype(np.int8)
    ix = flatnav.index.create("angular", 48, N, M, flatnav.data_type.DataType.int8)
    ix.set_num_threads(2)
    ix.add(X, 64, device=True, device_max_batch=1024)
    _check_graph(ix, N, M)
    d, l = ix.search(Q, K, 64)
    o = oracle_mod.OracleIndex.from_blob("angular", "int8", 48, N, N, M, np.asarray(ix._raw_blob()))
    od, ol = o.search(Q, K, 64)
    assert (ol == l).all() and (od == d).all()  # integer data: bit-exact, ties included


def test_c_abi_incremental_writes_equal_one_upload(oracle_mod):
    N, NQ, M, K = 8000, 400, 16, 10
    X, Q = ds.sift_like(N, NQ)
    o = oracle_mod.OracleIndex.create("l2", 128, N, M, "float32")
    o.add(X[: N // 2], 64)
    blob = np.asarray(o.blob())
    node_size, data_size = 128 * 4 + 4 * M + 4, 128 * 4
    half = N // 2
    whole = hip.DeviceIndex.upload(blob, node_size, data_size, M, half, "float32", "l2", 128)
    inc = hip.DeviceIndex.alloc(M, N, "float32", "l2", 128)  # room for N, holds N/2
    cut = 1234
    inc.write_nodes(0, blob[: cut * node_size], node_size, data_size)
    inc.write_nodes(cut, blob[cut * node_size: half * node_size], node_size, data_size)
    inc.set_live_nodes(half)
    dw, lw = whole.search(Q, K, 80)
    di, li = inc.search(Q, K, 80)
    assert (lw == li).all() and (dw == di).all()
    # rewrite some link rows: point node 7's row at 1..M, search must follow the new row like a fresh upload does
    rows = blob[: half * node_size].reshape(half, node_size).copy()
    new_row = np.arange(1, M + 1, dtype=np.uint32)
    rows[7, data_size:data_size + 4 * M] = new_row.view(np.uint8)
    inc.write_links(np.array([7], dtype=np.uint32), new_row[None, :])
    whole2 = hip.DeviceIndex.upload(rows.reshape(-1), node_size, data_size, M, half, "float32", "l2", 128)
    d2, l2 = whole2.search(Q, K, 80)
    di, li = inc.search(Q, K, 80)
    assert (l2 == li).all() and (d2 == di).all()
    inc.set_option("output_node_ids", 1)
    _, ids = inc.search(Q, K, 80)
    labels = rows[:, data_size + 4 * M:].copy().view(np.int32).reshape(-1)
    assert (labels[ids] == li).all()
    with pytest.raises(ValueError):
        inc.set_live_nodes(N + 1)
    with pytest.raises(RuntimeError):
        inc.write_nodes(N - 1, blob[: 2 * node_size], node_size, data_size)  # past capacity
    with pytest.raises(RuntimeError):
        inc.write_links(np.array([3], dtype=np.uint32), np.full((1, M), N + 5, dtype=np.uint32))  # id out of range


@pytest.mark.parametrize("metric,dt", [("l2", "float32"), ("angular", "float32"), ("l2", "uint8"), ("angular", "uint8"),
                                       ("l2", "int8"), ("angular", "int8")])
@pytest.mark.parametrize("spread", ["ties_everywhere", "few_ties"])
def test_sequential_device_insertion_reproduces_the_oracle_graph(flatnav, oracle_mod, metric, dt, spread):
    # Construction parity pinned (reference Index.h:353-378, 714-834): inserting ONE node per device batch is the
    # reference's sequential algorithm -- beam search over the nodes present, selectNeighbors to M/2, connectNeighbors
    # with first-free-slot / re-prune -- so on data whose distances are exact the graph must equal the oracle's
    # single-threaded build BYTE FOR BYTE, for all six index types, tie-heavy data included.
    rng = np.random.default_rng(21)
    N, dim, M, efc = 1200, 16, 8, 40
    lo, hi = (0, 4) if spread == "ties_everywhere" else (0, 60)
    if dt == "int8":
        lo, hi = lo - hi // 2, hi - hi // 2
    X = rng.integers(lo, hi, (N, dim)).astype(dt)
    labels = (rng.permutation(N) * 3 + 11).astype(np.int32)
    o = oracle_mod.OracleIndex.create(metric, dim, N, M, dt)
    o.add(X, efc, labels=labels)
    ix = flatnav.index.create(metric, dim, N, M, getattr(flatnav.data_type.DataType, dt))
    ix.add(X, efc, labels=labels.tolist(), device=True, device_max_batch=1, device_bootstrap=40)
    assert ix._cur_num_nodes == N
    want = np.asarray(o.blob())[: N * o.node_size].reshape(N, o.node_size)
    got = np.asarray(ix._raw_blob())[: N * o.node_size].reshape(N, o.node_size)
    bad = np.flatnonzero((want != got).any(axis=1))
    assert bad.size == 0, "first differing node %d of %d differing" % (bad[0], bad.size)


def test_sequential_device_insertion_on_random_shapes(flatnav, oracle_mod):
    # The same property over randomly drawn shapes: element type, metric, row width, link-row width, ef_construction,
    # tie density (FNV_FUZZ_TRIALS / FNV_FUZZ_SEED deepen the sweep).
    import os
    rng = np.random.default_rng(int(os.environ.get("FNV_FUZZ_SEED", "77")))
    for trial in range(int(os.environ.get("FNV_FUZZ_TRIALS", "16"))):
        dt = ["float32", "uint8", "int8"][trial % 3]
        metric = ["l2", "angular"][int(rng.integers(0, 2))]
        dim = int(rng.choice([4, 16, 33, 64, 128, 200]))
        M = int(rng.choice([2, 4, 8, 16, 32]))
        N = int(rng.integers(300, 1500))
        efc = int(rng.choice([5, 20, 40, 100]))
        hi = int(rng.choice([2, 4, 16, 100]))
        lo = -hi // 2 if dt == "int8" else 0
        X = rng.integers(lo, lo + hi, (N, dim)).astype(dt)
        o = oracle_mod.OracleIndex.create(metric, dim, N, M, dt)
        o.add(X, efc)
        ix = flatnav.index.create(metric, dim, N, M, getattr(flatnav.data_type.DataType, dt))
        boot = int(rng.choice([1, 8, 40]))
        if os.environ.get("FNV_FUZZ_BOOTSTRAP"):
            boot = int(os.environ["FNV_FUZZ_BOOTSTRAP"])
        # (odd trials: beam searches on the GPU, pruning / wiring by the host code -- the same sequential algorithm)
        ix.add(X, efc, device=True, device_max_batch=1, device_bootstrap=boot, device_wiring=trial % 2 == 0)
        want = np.asarray(o.blob())[: N * o.node_size].reshape(N, o.node_size)
        got = np.asarray(ix._raw_blob())[: N * o.node_size].reshape(N, o.node_size)
        bad = np.flatnonzero((want != got).any(axis=1))
        assert bad.size == 0, "trial %d (%s %s d=%d M=%d N=%d efc=%d hi=%d boot=%d): first differing node %d of %d differing" % (
            trial, dt, metric, dim, M, N, efc, hi, boot, bad[0], bad.size)


def test_batched_device_builds_on_random_shapes_are_valid_graphs(flatnav):
    # Batched insertion over randomly drawn shapes: every row holds distinct in-range neighbours (self id = empty slot),
    # every node is wired, the same call gives the same bytes, and the graph finds its own points.
    import os
    rng = np.random.default_rng(int(os.environ.get("FNV_FUZZ_SEED", "99")))
    for trial in range(int(os.environ.get("FNV_FUZZ_TRIALS", "8"))):
        dt = ["float32", "uint8", "int8"][trial % 3]
        metric = "l2" if trial % 2 == 0 else "angular"
        dim = int(rng.choice([8, 32, 100, 128]))
        M = int(rng.choice([4, 8, 16, 32, 48]))
        N = int(rng.integers(3000, 20000))
        efc = int(rng.choice([20, 64, 100]))
        batch = int(rng.choice([7, 256, 4096]))
        boot = int(rng.choice([64, 2048]))
        hi = int(rng.choice([4, 16, 100]))
        lo = -hi // 2 if dt == "int8" else 0
        X = rng.integers(lo, lo + hi, (N, dim)).astype(dt)
        what = "trial %d: %s %s d=%d M=%d N=%d efc=%d batch=%d boot=%d hi=%d" % (trial, dt, metric, dim, M, N, efc, batch, boot, hi)
        blobs = []
        for _ in range(2 if trial % 4 == 0 else 1):
            ix = flatnav.index.create(metric, dim, N, M, getattr(flatnav.data_type.DataType, dt))
            ix.set_num_threads(4)
            ix.add(X, efc, device=True, device_max_batch=batch, device_bootstrap=boot)
            blobs.append(np.asarray(ix._raw_blob()).copy())
        assert len(blobs) == 1 or np.array_equal(blobs[0], blobs[1]), what
        _check_graph(ix, N, M)
        if metric == "l2" and hi >= 16 and M >= 8 and boot >= 2048 and trial % 3 != 2:  # how often does a node find itself,
            host = flatnav.index.create(metric, dim, N, M, getattr(flatnav.data_type.DataType, dt))  # against the host builder
            host.set_num_threads(4)
            host.add(X, efc)
            found = [float((g.search(X[:300], 1, 64)[0][:, 0] == 0).mean()) for g in (ix, host)]
            assert found[0] > found[1] - 0.1, what + " self-recall device %.2f host %.2f" % tuple(found)


def test_batched_device_build_is_deterministic(flatnav):
    # Same data, same options -> same bytes, run after run (requests are grouped by a stable sort, not by arrival).
    N, M = 30000, 32
    X, _ = ds.sift_like(N, 10)
    blobs = []
    for _ in range(2):
        ix = flatnav.index.create("l2", 128, N, M)
        ix.set_num_threads(4)
        ix.add(X, 100, device=True, device_max_batch=4096)
        blobs.append(np.asarray(ix._raw_blob()).copy())
    assert np.array_equal(blobs[0], blobs[1])
