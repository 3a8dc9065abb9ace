"""Round 6 on the GPU:
  * a reordered index (gorder / rcm: the reference's permutation, tests/test_reference_pins.py) is searched by the GPU exactly as
    the oracle searches the relabelled node store -- a test of its own for all six index types (before, only the
    random operation sequences of test_gpu_python_api.py reached `reorder`, when the draw happened to pick it);
  * replicas inherit what the source measured (fnv_replica_refresh / fnv_search_batch_multi) -- no exploratory launch on a
    replica after one fnv_tune on the source, identical bytes."""
import numpy as np
import pytest

from flatnav_amd import datasets as ds

pytestmark = pytest.mark.gpu

CASES = [("l2", "float32"), ("angular", "float32"), ("l2", "uint8"), ("angular", "uint8"), ("l2", "int8"), ("angular", "int8")]


@pytest.fixture(scope="module")
def flatnav():
    import flatnav_amd

    return flatnav_amd


@pytest.mark.parametrize("metric,dt", CASES, ids=["%s-%s" % c for c in CASES])
def test_reordered_index_is_searched_like_the_oracle(flatnav, oracle_mod, metric, dt):
    rng = np.random.default_rng(66)
    N, dim, M = 6000, 48, 16
    lo, hi = (-20, 20) if dt == "int8" else (0, 40)
    X = rng.integers(lo, hi, (N, dim)).astype(dt)
    Q = rng.integers(lo, hi, (500, dim)).astype(dt)
    ix = flatnav.index.create(metric, dim, N, M, getattr(flatnav.data_type.DataType, dt))
    ix.set_num_threads(4)
    ix.add(X, 48, labels=[int(v) for v in rng.permutation(N) * 3 + 11])
    before = ix.search(Q, 10, 80)  # (the device mirror exists before the relabelling and has to follow it)
    for methods in (["gorder"], ["rcm"], ["gorder", "rcm"]):
        ix.reorder(methods)
        o = oracle_mod.OracleIndex.from_blob("l2" if metric == "l2" else "ip", dt, dim, N, N, M, np.asarray(ix._raw_blob()))
        for K, ef in ((10, 80), (1, 16), (25, 200)):
            od, ol = o.search(Q, K, ef)
            gd, gl = ix.search(Q, K, ef)
            assert np.array_equal(gl, ol) and np.array_equal(gd.view(np.uint32), od.view(np.uint32)), (methods, K, ef)
        sd, sl = ix.search_single(Q[7], 10, 80)
        od, ol = o.search(Q[7:8], 10, 80)
        assert np.array_equal(np.asarray(sl).ravel(), ol[0]) and np.array_equal(np.asarray(sd, dtype=np.float32).ravel(), od[0])
    # a relabelling moves the entry-scan sample, so answers may differ from before -- but labels travel with
    # their nodes: every returned label is still one of the index's labels
    after = ix.search(Q, 10, 80)
    valid = set(int(v) for v in np.asarray(ix._raw_blob()).reshape(N, -1)[:, -4:].copy().view(np.int32).ravel())
    assert set(after[1].ravel().tolist()) <= valid and set(before[1].ravel().tolist()) <= valid


def test_replicas_inherit_the_sources_measurements(oracle_mod):
    from flatnav_amd import hip

    X, Q = ds.sift_like(30000, 12000)
    o = oracle_mod.OracleIndex.create("l2", 128, 30000, 16)
    o.add(X, 48)
    src = hip.DeviceIndex.upload(o.blob(), o.node_size, o.data_size, o.M, o.cur_nodes, "float32", "l2", 128)
    want = o.search(Q, 10, 64, stats=True, threads=8)
    # (1) tuned BEFORE replication: fnv_replicate -> fnv_replica_refresh hands the measurements over
    src.tune(Q[:6000], 10, 64)
    reps = src.replicate([0, 0])
    for r in reps:
        got = r.search(Q[:6000], 10, 64, stats=True)
        assert not r.launch_info()["exploratory"], r.launch_info()
        assert np.array_equal(got[1], want[1][:6000]) and np.array_equal(got[0].view(np.uint32), want[0][:6000].view(np.uint32))
        assert np.array_equal(got[2]["n_dist"], want[2]["n_dist"][:6000]) and np.array_equal(got[2]["n_hops"], want[2]["n_hops"][:6000])
    # (2) another beam width tuned AFTER replication: the multi-handle call brings the replicas up to date
    src.tune(Q[:4000], 10, 100)
    want100 = o.search(Q, 10, 100, stats=True, threads=8)
    got = hip.search_multi([src] + reps, Q, 10, 100, stats=True)
    assert np.array_equal(got[1], want100[1]) and np.array_equal(got[0].view(np.uint32), want100[0].view(np.uint32))
    assert all(np.array_equal(got[2][k], want100[2][k]) for k in ("n_dist", "n_hops"))
    infos = [h.launch_info() for h in [src] + reps]
    assert not any(i["exploratory"] for i in infos), infos
    # (3) an option that changes the plan on the source alone: its measurements are void, the replicas keep answering with
    # their own (old) options until the next refresh and never take measurements made under other options
    src.set_option("visited_factor", 20)
    got = hip.search_multi([src] + reps, Q, 10, 100, stats=True)
    assert np.array_equal(got[1], want100[1]) and np.array_equal(got[0].view(np.uint32), want100[0].view(np.uint32))
    src.refresh_replicas(reps)
    got = hip.search_multi([src] + reps, Q, 10, 100, stats=True)
    assert np.array_equal(got[1], want100[1]) and all(np.array_equal(got[2][k], want100[2][k]) for k in ("n_dist", "n_hops"))


# ---- SPLIT ROWS (csrc/distance.hpp, beam_search.hip row_layout): three whole lines in the table + the last 16 / 32 bytes of every
#      row in a dense side table.  Search parity over the dims that split: tests/test_gpu_round3.py::test_rows_on_whole_lines_keep_every_bit.
SPLIT_CASES = [("l2", "float32", 100), ("angular", "float32", 104), ("l2", "uint8", 400), ("angular", "int8", 410), ("l2", "float32", 97)]


@pytest.mark.parametrize("metric,dt,dim", SPLIT_CASES, ids=["%s-%s-%d" % c for c in SPLIT_CASES])
def test_split_rows_device_builder_reproduces_the_oracle_graph(flatnav, oracle_mod, metric, dt, dim):
    # the device builder on split rows: the new nodes' vectors become queries (two strided copies), the wiring kernels stage a
    # node's vector from both tables -- ONE node per batch must reproduce the oracle's single-threaded graph byte for byte
    # (reference Index.h:353-378, 714-834), and the batched builder's graph must answer like the oracle searching it
    import ctypes

    from flatnav_amd import hip

    rng = np.random.default_rng(dim)
    N, M, efc = 1500, 8, 40
    lo, hi = (-6, 6) if dt == "int8" else (0, 12)
    X = rng.integers(lo, hi, (N, dim)).astype(dt)
    Q = rng.integers(lo, hi, (300, dim)).astype(dt)
    o = oracle_mod.OracleIndex.create(metric, dim, N, M, dt)
    o.add(X, efc)
    ix = flatnav.index.create(metric, dim, N, M, getattr(flatnav.data_type.DataType, dt))
    ix.add(X, efc, device=True, device_max_batch=1, device_bootstrap=40)
    dev = hip.DeviceIndex(ctypes.c_void_p(ix.device_handle()), owned=False)
    assert dev.row_bytes == 384 and dev.tail_bytes in (16, 32)
    want = np.asarray(o.blob())[: N * o.node_size].reshape(N, o.node_size)
    got = np.asarray(ix._raw_blob())[: N * o.node_size].reshape(N, o.node_size)
    bad = np.flatnonzero((want != got).any(axis=1))
    assert bad.size == 0, "first differing node %d of %d differing" % (bad[0], bad.size)
    for K, ef in ((10, 64), (1, 8), (20, 300)):  # beams in registers (one / two chunks) and in LDS
        od, ol, ost = o.search(Q, K, ef, stats=True)
        gd, gl, gst = dev.search(Q, K, ef, stats=True)
        assert np.array_equal(gl, ol) and np.array_equal(gd.view(np.uint32), od.view(np.uint32)), (K, ef)
        assert np.array_equal(gst["n_dist"], ost["n_dist"]) and np.array_equal(gst["n_hops"], ost["n_hops"])
    dev.set_option("sorted_beam", 0)  # the two-heap kernel
    gd, gl = dev.search(Q, 10, 64)
    od, ol = o.search(Q, 10, 64)
    assert np.array_equal(gl, ol) and np.array_equal(gd.view(np.uint32), od.view(np.uint32))
    dev.set_option("sorted_beam", 2)
    dev.set_option("entry_kernel", 1)  # K0 holds one table's rows in its LDS tiles: ignored on split rows, same answers
    gd, gl = dev.search(Q, 10, 64)
    assert np.array_equal(gl, ol) and np.array_equal(gd.view(np.uint32), od.view(np.uint32))
    # batched insertion (the bench's builder): a valid graph that GPU and oracle search identically
    big = flatnav.index.create(metric, dim, 6000, 16, getattr(flatnav.data_type.DataType, dt))
    Xb = rng.integers(lo, hi, (6000, dim)).astype(dt)
    big.add(Xb[:3500], 48, device=True)
    big.add(Xb[3500:], 48, labels=list(range(3500, 6000)), device=True)  # the index grows: later rows of both tables
    ob = oracle_mod.OracleIndex.from_blob("l2" if metric == "l2" else "ip", dt, dim, 6000, 6000, 16, np.asarray(big._raw_blob()))
    od, ol = ob.search(Q, 10, 80)
    gd, gl = big.search(Q, 10, 80)
    assert np.array_equal(gl, ol) and np.array_equal(gd.view(np.uint32), od.view(np.uint32))


def test_split_rows_incremental_writes_views_adoption_and_replicas(oracle_mod):
    from flatnav_amd import hip

    rng = np.random.default_rng(100)
    N, NQ, M, K, dim = 8000, 400, 16, 10, 100
    X = rng.integers(0, 30, (N, dim)).astype(np.float32)
    Q = rng.integers(0, 30, (NQ, dim)).astype(np.float32)
    o = oracle_mod.OracleIndex.create("l2", dim, N, M, "float32")
    o.add(X[: N // 2], 64)
    blob = np.asarray(o.blob())
    node_size, data_size, half = dim * 4 + 4 * M + 4, dim * 4, N // 2
    want = o.search(Q, K, 80, stats=True)
    whole = hip.DeviceIndex.upload(blob, node_size, data_size, M, half, "float32", "l2", dim)
    inc = hip.DeviceIndex.alloc(M, N, "float32", "l2", dim)  # room for N, holds N / 2: its side table starts at N * 384
    assert whole.tail_bytes == inc.tail_bytes == 16 and whole.row_bytes == inc.row_bytes == 384
    inc.write_nodes(0, blob[: 1234 * node_size], node_size, data_size)
    inc.write_nodes(1234, blob[1234 * node_size: half * node_size], node_size, data_size)
    inc.set_live_nodes(half)
    for dev in (whole, inc):
        got = dev.search(Q, K, 80, stats=True)
        assert np.array_equal(got[1], want[1]) and np.array_equal(got[0].view(np.uint32), want[0].view(np.uint32))
        assert np.array_equal(got[2]["n_dist"], want[2]["n_dist"])
    view = inc.view()
    guest = hip.DeviceIndex.adopt(inc.device_buffers(), M, N, "float32", "l2", dim, keep_alive=inc)  # (capacity N: the owner's layout)
    guest.set_live_nodes(half)
    assert guest.tail_bytes == 16
    for dev in (view, guest):
        got = dev.search(Q, K, 80)
        assert np.array_equal(got[1], want[1]) and np.array_equal(got[0].view(np.uint32), want[0].view(np.uint32))
    view.close()
    guest.close()
    # replicas: both tables travel (the live rows of each), a refresh after growth as well
    reps = whole.replicate([0, 0])
    got = hip.search_multi([whole] + reps, Q, K, 80, stats=True)
    assert np.array_equal(got[1], want[1]) and np.array_equal(got[0].view(np.uint32), want[0].view(np.uint32))
    grown = inc.replicate([0])
    o.add(X[N // 2:], 64)
    blob2 = np.asarray(o.blob())
    inc.write_nodes(0, blob2, node_size, data_size)  # (links of old nodes changed too)
    inc.set_live_nodes(N)
    inc.refresh_replicas(grown)
    want2 = o.search(Q, K, 80)
    for dev in (inc, grown[0]):
        got = dev.search(Q, K, 80)
        assert np.array_equal(got[1], want2[1]) and np.array_equal(got[0].view(np.uint32), want2[0].view(np.uint32))


def test_advice_r5_option_bounds_and_launch_records_from_the_serving_lane(oracle_mod):
    # "tie_log_entries" is bounded (it sizes max_slots x entries x 8 bytes of workspace); after concurrent callers
    # -- some served by hidden lanes -- the handle's "most recent launch" records (replayed queries, hand-over statistics,
    # kernel time) all describe ONE launch: the one of the lane that served last
    import threading

    from flatnav_amd import hip

    X, Q = ds.sift_like(20000, 6000)
    Xu, Qu = X.astype(np.uint8), Q.astype(np.uint8)
    o = oracle_mod.OracleIndex.create("l2", 128, 20000, 32, "uint8")
    o.add(Xu, 64)
    dev = hip.DeviceIndex.upload(o.blob(), o.node_size, o.data_size, o.M, o.cur_nodes, "uint8", "l2", 128)
    with pytest.raises(ValueError):
        dev.set_option("tie_log_entries", (1 << 20) + 1)
    dev.set_option("tie_log_entries", 1 << 20)  # the largest allowed: 8 MB per slot -- a small launch still fits
    want = o.search(Qu, 10, 52, stats=True, threads=8)
    got = dev.search(Qu[:64], 10, 52, stats=True)
    assert np.array_equal(got[1], want[1][:64]) and np.array_equal(got[2]["n_dist"], want[2]["n_dist"][:64])
    dev.set_option("tie_log_entries", 0)
    dev.set_option("sorted_beam", 1)
    dev.set_option("sorted_variant", 1)  # every query through the merged-beam kernel: tied ones are handed over
    outs = [None] * 4

    def work(t):
        for rep in range(3):
            outs[t] = dev.search(Qu[t * 1500:(t + 1) * 1500], 10, 52, stats=True)

    for _ in range(3):
        th = [threading.Thread(target=work, args=(t,)) for t in range(4)]
        [t.start() for t in th]
        [t.join() for t in th]
        for t in range(4):
            assert np.array_equal(outs[t][1], want[1][t * 1500:(t + 1) * 1500])
            assert np.array_equal(outs[t][0].view(np.uint32), want[0][t * 1500:(t + 1) * 1500].view(np.uint32))
            assert np.array_equal(outs[t][2]["n_hops"], want[2]["n_hops"][t * 1500:(t + 1) * 1500])
        r, h = dev.replayed_queries(), dev.handover_stats()
        assert h["resumed"] + h["from_scratch"] == r["total"], (r, h)  # (both from the same launch)
        assert dev.last_kernel_ms() > 0
    assert sum(1 for w in hip.lane_workspaces(dev)[1:] if w) >= 1  # (lanes really served)


def test_zero_copy_host_searches_change_no_byte(oracle_mod):
    # "host_zero_copy" (default: on for every call that fits the pinned staging buffer): the kernel reads the queries from pinned
    # host memory and writes results, counters and the error flag straight into it.  The oracle's bytes with it and without it,
    # from 1 to 1500 queries (beyond the buffer the call copies as before), K > results (fewer than K reachable: -1 / inf
    # padding and the count), concurrent callers on hidden lanes, and the error flag: a candidate heap that overflows its
    # spill area must still raise through the pinned copy of the flag.
    import threading

    from flatnav_amd import hip

    X, Q = ds.sift_like(20000, 1500)
    Xu, Qu = X.astype(np.uint8), Q.astype(np.uint8)
    o = oracle_mod.OracleIndex.create("l2", 128, 20000, 32, "uint8")
    o.add(Xu, 64)
    dev = hip.DeviceIndex.upload(o.blob(), o.node_size, o.data_size, o.M, o.cur_nodes, "uint8", "l2", 128)
    for K, ef in ((10, 52), (1, 8), (40, 300)):
        want = o.search(Qu, K, ef, stats=True, threads=8)
        for nq in (1, 3, 64, 700, 1500):
            for zc in (1 << 20, 0, 2):  # 2: only the smallest batches
                dev.set_option("host_zero_copy", zc)
                got = dev.search(Qu[:nq], K, ef, stats=True)
                what = "K=%d ef=%d %d queries host_zero_copy=%d" % (K, ef, nq, zc)
                assert np.array_equal(got[1], want[1][:nq]) and np.array_equal(got[0].view(np.uint32), want[0][:nq].view(np.uint32)), what
                assert all(np.array_equal(got[2][k], want[2][k][:nq]) for k in ("count", "n_dist", "n_hops")), what
    dev.set_option("host_zero_copy", 1 << 20)
    want = o.search(Qu, 10, 52, stats=True, threads=8)
    outs, errs = [None] * 6, []

    def work(t):
        try:
            for rep in range(20):
                q0 = (t * 37 + rep * 11) % 1400
                n = 1 + (t + rep) % 5
                d, l = dev.search(Qu[q0:q0 + n], 10, 52)
                assert np.array_equal(l, want[1][q0:q0 + n]) and np.array_equal(d.view(np.uint32), want[0][q0:q0 + n].view(np.uint32))
        except Exception as exc:  # noqa: BLE001
            errs.append(repr(exc))

    th = [threading.Thread(target=work, args=(t,)) for t in range(6)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    # fewer than K reachable results: a tiny graph, K larger than it
    small = oracle_mod.OracleIndex.create("l2", 128, 40, 4, "uint8")
    small.add(Xu[:40], 8)
    ds_ = hip.DeviceIndex.upload(small.blob(), small.node_size, small.data_size, small.M, small.cur_nodes, "uint8", "l2", 128)
    ws = small.search(Qu[:5], 60, 80, stats=True)
    gs = ds_.search(Qu[:5], 60, 80, stats=True)
    assert np.array_equal(gs[1], ws[1]) and np.array_equal(gs[2]["count"], ws[2]["count"]) and (gs[2]["count"] < 60).all()
    # the error flag travels through the pinned slab: a candidate heap with no LDS home and a one-entry spill area overflows
    # (the configuration tests/test_gpu_parity.py uses for the same error on a copied call)
    dev.set_option("sorted_beam", 0)
    dev.set_option("cand_slots", 8)
    dev.set_option("spill_entries", 1)
    with pytest.raises(RuntimeError, match="spill"):
        dev.search(Qu[:100], 10, 100)
    dev.set_option("host_zero_copy", 0)
    with pytest.raises(RuntimeError, match="spill"):
        dev.search(Qu[:100], 10, 100)


def test_callers_with_pinned_arrays_run_zero_copy_at_any_batch_size(oracle_mod):
    # fnv_search_batch on arrays that are ALREADY pinned host memory (torch pin_memory): beyond the staging buffer's size the kernel
    # reads the queries from and writes the results into the caller's own memory; mixed (pinned queries, pageable outputs) and
    # pageable callers copy as before.  The oracle's bytes on every path.
    import torch

    from flatnav_amd import hip

    X, Q = ds.sift_like(20000, 6000)  # 6000 x 512 bytes = 3 MB of queries: three times the staging buffer
    o = oracle_mod.OracleIndex.create("l2", 128, 20000, 32)
    o.add(X, 64)
    dev = hip.DeviceIndex.upload(o.blob(), o.node_size, o.data_size, o.M, o.cur_nodes, "float32", "l2", 128)
    want = o.search(Q, 10, 52, threads=8)
    qpin = torch.from_numpy(Q).pin_memory()
    dpin, lpin = torch.empty((6000, 10), dtype=torch.float32).pin_memory(), torch.empty((6000, 10), dtype=torch.int32).pin_memory()
    for rep in range(3):
        dpin.zero_(); lpin.zero_()
        dev.search_into(qpin.numpy(), 10, 52, dpin.numpy(), lpin.numpy())
        assert np.array_equal(lpin.numpy(), want[1]) and np.array_equal(dpin.numpy().view(np.uint32), want[0].view(np.uint32))
    d2, l2 = np.empty((6000, 10), np.float32), np.empty((6000, 10), np.int32)  # pageable outputs: the call copies
    dev.search_into(qpin.numpy(), 10, 52, d2, l2)
    assert np.array_equal(l2, want[1]) and np.array_equal(d2.view(np.uint32), want[0].view(np.uint32))
    dev.search_into(Q, 10, 52, dpin.numpy(), lpin.numpy())  # pageable queries, pinned outputs
    assert np.array_equal(lpin.numpy(), want[1])
    dev.set_option("host_zero_copy", 0)  # off: pinned callers copy too
    dpin.zero_()
    dev.search_into(qpin.numpy(), 10, 52, dpin.numpy(), lpin.numpy())
    assert np.array_equal(lpin.numpy(), want[1]) and np.array_equal(dpin.numpy().view(np.uint32), want[0].view(np.uint32))
    with pytest.raises(ValueError):
        dev.search_into(qpin.numpy(), 10, 52, dpin.numpy()[:, :5], lpin.numpy())
