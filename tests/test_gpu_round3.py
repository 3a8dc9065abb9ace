"""Round-3 additions to the C ABI, each against the oracle or against the library's own exact path:
  * row stride on whole 128-byte lines (FLATNAV_ROW_PAD_PCT): every bit of every result unchanged;
  * fnv_tune / "sorted_variant" / fnv_last_launch_info: the adaptive kernel choice settled in one call, no exploratory
    launch afterwards, every pinned variant returns the oracle's bytes;
  * views count on their source (free refused, growth followed); replicas take over the source's options;
  * fnv_search_batch_multi drives every shard from its own host thread: the second handle's launch is enqueued before the
    first one's search completes;
  * fnv_gather_ceiling returns a plausible rate."""
import os

import numpy as np
import pytest

from flatnav_amd import datasets as ds

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hipmod():
    from flatnav_amd import hip

    assert hip.device_count() >= 1, "no MI355X visible"
    return hip


def _upload(hipmod, ix):
    return hipmod.DeviceIndex.upload(ix.blob(), ix.node_size, ix.data_size, ix.M, ix.cur_nodes, ix.dtype, ix.metric, ix.dim)


@pytest.mark.parametrize("dt,dim,metric", [("float32", 100, "angular"), ("uint8", 100, "l2"), ("float32", 200, "l2"),
                                           ("int8", 37, "angular"), ("float32", 760, "l2"),
                                           # round 6, SPLIT ROWS (three whole lines + a 16 / 32-byte tail in a side table):
                                           ("float32", 104, "l2"), ("float32", 97, "l2"), ("uint8", 400, "l2"), ("int8", 410, "angular"),
                                           ("float32", 96, "angular"), ("uint8", 384, "l2")])  # ... and plain three-line rows
def test_rows_on_whole_lines_keep_every_bit(oracle_mod, hipmod, dt, dim, metric, monkeypatch):
    rng = np.random.default_rng(dim)
    N, NQ = 6000, 400
    if dt == "float32":
        X = rng.standard_normal((N, dim), dtype=np.float32)
        Q = rng.standard_normal((NQ, dim), dtype=np.float32)
        if metric == "angular":
            X /= np.linalg.norm(X, axis=1, keepdims=True)
            Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    else:
        lo, hi = (0, 40) if dt == "uint8" else (-20, 20)
        X = rng.integers(lo, hi, (N, dim)).astype(dt)
        Q = rng.integers(lo, hi, (NQ, dim)).astype(dt)
    ix = oracle_mod.OracleIndex.create(metric, dim, N, 16, dt)
    ix.add(X, 48)
    esize = 4 if dt == "float32" else 1
    rb16 = (dim * esize + 15) // 16 * 16
    rb128 = (rb16 + 127) // 128 * 128
    monkeypatch.setenv("FLATNAV_SPLIT_ROWS", "0")  # (the padding rules of rounds 2-5 first)
    monkeypatch.setenv("FLATNAV_ROW_PAD_PCT", "0")
    plain = _upload(hipmod, ix)
    assert plain.row_bytes == rb16 and plain.tail_bytes == 0
    monkeypatch.delenv("FLATNAV_ROW_PAD_PCT")
    padded = _upload(hipmod, ix)
    assert padded.row_bytes == (rb128 if (rb128 - rb16) * 100 <= 30 * rb16 else rb16) and padded.tail_bytes == 0
    monkeypatch.setenv("FLATNAV_ROW_PAD_PCT", "400")
    always = _upload(hipmod, ix)
    assert always.row_bytes == rb128
    monkeypatch.delenv("FLATNAV_ROW_PAD_PCT")
    monkeypatch.delenv("FLATNAV_SPLIT_ROWS")
    layouts = [plain, padded, always]
    rem = rb16 % 128
    splits = rb16 - rem == 384 and 0 < rem <= 32  # csrc/beam_search.hip row_layout (the side table of 6000 rows is tiny)
    default = _upload(hipmod, ix)  # what a caller gets
    if splits:
        assert default.row_bytes == 384 and default.tail_bytes == rem
        assert default.device_buffers()[0][1] == N * (384 + rem)
        layouts.append(default)
        monkeypatch.setenv("FLATNAV_SPLIT_TAIL_MAX_MB", "0")  # a side table that would not stay cached: rows are padded instead
        assert _upload(hipmod, ix).tail_bytes == 0
        monkeypatch.delenv("FLATNAV_SPLIT_TAIL_MAX_MB")
    else:
        assert default.row_bytes == padded.row_bytes and default.tail_bytes == 0
    # (split rows keep the lane-to-chunk mapping of the four-line padded row -- lane g: chunks g, g + 8, g + 16, g + 24 -- but not
    #  that of the 16-byte stride's clamped path)
    cfg = [pick_cfg(dev.row_bytes // 16, dev.tail_bytes // 16) for dev in layouts]
    for ef in (30, 150):
        for mode in (2, 0):
            outs = []
            for dev in layouts:
                dev.set_option("sorted_beam", mode)
                outs.append(dev.search(Q, 10, ef, stats=True))
            for i in range(1, len(layouts)):
                other = outs[i]
                if i == 3:  # split rows == the padded four-line row, bit for bit, floats included
                    assert np.array_equal(outs[1][0].view(np.uint32), other[0].view(np.uint32)) and np.array_equal(outs[1][1], other[1])
                    assert all(np.array_equal(outs[1][2][k], other[2][k]) for k in ("count", "n_dist", "n_hops"))
                if dt != "float32" or cfg[i] == cfg[0]:
                    # integer arithmetic, or the same lane-to-chunk mapping (the zero padding adds nothing to either
                    # partial sum): the same bits
                    assert np.array_equal(outs[0][0].view(np.uint32), other[0].view(np.uint32))
                    assert np.array_equal(outs[0][1], other[1])
                    assert all(np.array_equal(outs[0][2][k], other[2][k]) for k in ("count", "n_dist", "n_hops"))
                else:  # another row configuration sums the same products in another order: float tolerance
                    same = (outs[0][1] == other[1]).all(axis=1)
                    assert same.mean() >= 0.999 and np.allclose(outs[0][0][same], other[0][same], rtol=1e-5, atol=1e-6)
    if dt != "float32":  # integer data: also bit-exact against the oracle in every layout
        od, ol, ost = ix.search(Q, 10, 150, stats=True)
        for dev in layouts:
            gd, gl, gst = dev.search(Q, 10, 150, stats=True)
            assert np.array_equal(ol, gl) and np.array_equal(od.view(np.uint32), gd.view(np.uint32))
            assert np.array_equal(ost["n_dist"], gst["n_dist"])


def pick_cfg(nchunks, tail_chunks=0):
    """csrc/kernel_table.h pick_row_cfg."""
    cfgs = [8, 16, 32, 64, 128, 256]
    if tail_chunks or nchunks == 24:  # three whole lines (+ a side-table tail: split rows, round 6)
        return 7
    if nchunks == 192:  # rows of exactly 3 KB: every lane loads its three chunks, the query lives in registers (round 4)
        return 6
    for c, span in enumerate(cfgs):
        if span >= nchunks:
            return c
    return 5


def test_tune_settles_the_kernel_choice_and_variants_can_be_pinned(oracle_mod, hipmod):
    X, Q = ds.sift_like(30000, 6000)  # integer-valued: ties -> the variants differ in speed, never in results
    ix = oracle_mod.OracleIndex.create("l2", 128, 30000, 16)
    ix.add(X, 64)
    want = ix.search(Q, 10, 64, stats=True)
    dev = _upload(hipmod, ix)
    import torch

    dq = torch.from_numpy(Q).cuda()
    od = torch.empty((len(Q), 10), dtype=torch.float32, device="cuda")
    ol = torch.empty((len(Q), 10), dtype=torch.int32, device="cuda")
    ond = torch.zeros(len(Q), dtype=torch.int64, device="cuda")

    def launch(ef):
        dev.search_device(dq.data_ptr(), len(Q), 10, ef, 100, od.data_ptr(), ol.data_ptr(), 0, ond.data_ptr())
        torch.cuda.synchronize()
        dev.status()
        return dev.launch_info()

    # without tuning, the first big launches of a beam width are exploratory samples
    assert launch(64)["exploratory"]
    dev.set_option("sorted_beam", 2)  # resets what has been measured
    dev.tune(Q, 10, 64)
    finals = set()
    for _ in range(6):
        info = launch(64)
        assert not info["exploratory"]
        finals.add(info["variant_id"])
        assert np.array_equal(want[1], ol.cpu().numpy()) and np.array_equal(want[0].view(np.uint32), od.cpu().numpy().view(np.uint32))
        assert np.array_equal(want[2]["n_dist"], ond.cpu().numpy().astype(np.uint64))
    assert len(finals) == 1  # the first launch after fnv_tune already runs the final variant, and so does every later one
    # another beam width has not been measured yet
    assert launch(40)["exploratory"]
    # device-resident queries
    dev.tune(int(dq.data_ptr()), 10, 40, nq=len(Q))
    assert not launch(40)["exploratory"]
    dev.search(Q, 10, 40)  # the host-buffer entry point launches the same way
    assert not dev.launch_info()["exploratory"]
    # every variant pinned: same bytes as the oracle
    slots = dev.launch_geometry()["blocks_per_cu"] * 256
    for v in range(7):  # (6, round 4: the merged-beam kernel for every query + exact shadows of the last ones on idle slots)
        dev.set_option("sorted_variant", v)
        got = dev.search(Q, 10, 64, stats=True)
        info = dev.launch_info()
        assert not info["exploratory"] and (info["variant_id"] == v or (2 <= v <= 5 and len(Q) <= slots))
        assert np.array_equal(want[1], got[1]) and np.array_equal(want[0].view(np.uint32), got[0].view(np.uint32))
        assert all(np.array_equal(want[2][k], got[2][k]) for k in ("count", "n_dist", "n_hops"))
    with pytest.raises(ValueError):
        dev.set_option("sorted_variant", 7)
    with pytest.raises(ValueError):
        dev.tune(Q[:0], 10, 64)


def test_tune_measures_the_lds_layout_and_never_changes_results(oracle_mod, hipmod):
    # fnv_tune also picks the per-query LDS layout (heap home, visited-table size) by measurement: whatever it picks,
    # ids, distances and counters stay the oracle's
    X, Q = ds.lowrank_normalized(40000, 4096, dim=100, rank=24, seed=100)
    ix = oracle_mod.OracleIndex.create("angular", 100, 40000, 32)
    ix.add(X, 64)
    dev = _upload(hipmod, ix)
    for ef in (64, 128, 300):
        before = dev.search(Q, 10, ef, stats=True)
        g0 = dev.launch_geometry()
        dev.tune(Q, 10, ef)
        after = dev.search(Q, 10, ef, stats=True)
        g1 = dev.launch_geometry()
        print("ef=%d: rules %s -> tuned %s, %s" % (ef, {k: g0[k] for k in ("blocks_per_cu", "visited_slots", "cand_slots")},
                                                 {k: g1[k] for k in ("blocks_per_cu", "visited_slots", "cand_slots")},
                                                 dev.launch_info()["variant"]))
        assert not dev.launch_info()["exploratory"]
        assert np.array_equal(before[1], after[1]) and np.array_equal(before[0].view(np.uint32), after[0].view(np.uint32))
        assert all(np.array_equal(before[2][k], after[2][k]) for k in ("count", "n_dist", "n_hops"))
    # an option set by the caller wins over the measurement, and resets it
    dev.set_option("visited_slots", 2048)
    dev.tune(Q, 10, 128)
    dev.search(Q, 10, 128)
    assert dev.launch_geometry()["visited_slots"] == 2048


@pytest.mark.parametrize("case", ["u8_every_query_ties", "f32_integer_valued", "f32_real"])
def test_small_launches_run_exact_shadows(oracle_mod, hipmod, case):
    # "shadow_exact": a launch that fills at most a quarter of the slots runs, next to the merged-beam search of every
    # query, an exact search of the same query -- answers stay the oracle's, tie or no tie, shadow or no shadow
    rng = np.random.default_rng(11)
    if case == "u8_every_query_ties":
        X = rng.integers(0, 4, (6000, 16)).astype(np.uint8)
        Q = rng.integers(0, 4, (700, 16)).astype(np.uint8)
        dt, metric = "uint8", "l2"
    elif case == "f32_integer_valued":
        X, Q = ds.sift_like(8000, 700)
        dt, metric = "float32", "l2"
    else:
        X, Q = ds.lowrank_normalized(8000, 700, dim=100, rank=24, seed=100)
        dt, metric = "float32", "angular"
    ix = oracle_mod.OracleIndex.create(metric, X.shape[1], len(X), 16, dt)
    ix.add(X, 48)
    dev = _upload(hipmod, ix)
    exact_ids = case != "f32_real"
    for ef, K in ((40, 10), (150, 25)):
        want = ix.search(Q, K, ef, stats=True)
        for nq in (1, 7, 64, 700):
            for mode in (1, 0):
                dev.set_option("shadow_exact", mode)
                got = dev.search(Q[:nq], K, ef, stats=True)
                info = dev.launch_info()
                # (a quarter of the slots a CU REALLY keeps resident -- LDS comes in 1280-byte granules, round 4 -- times 256 CUs:
                #  700 queries at ef=150 are just over it on some layouts)
                small = 4 * nq <= dev.launch_geometry()["blocks_per_cu"] * 256
                assert small or nq == 700
                assert info["shadow"] == (bool(mode) and small), (nq, mode, info, dev.launch_geometry())
                if exact_ids:
                    assert np.array_equal(want[1][:nq], got[1]) and np.array_equal(want[0][:nq].view(np.uint32), got[0].view(np.uint32))
                    assert all(np.array_equal(want[2][k][:nq], got[2][k]) for k in ("count", "n_dist", "n_hops")), (case, nq, mode)
                else:
                    same = (want[1][:nq] == got[1]).all(axis=1)  # (the survey's float bar, as everywhere: 99.9 %)
                    assert same.mean() >= 0.999 and np.allclose(want[0][:nq][same], got[0][same], rtol=1e-5, atol=1e-6)
        if case == "u8_every_query_ties":
            dev.set_option("shadow_exact", 1)
            dev.search(Q[:64], K, ef)
            assert dev.replayed_queries()["total"] > 30  # the merged-beam pass gave most of them up: their shadows answered
    # repeated small launches leave the per-slot state clean (stopped shadows hand their HBM bitmap back zeroed)
    dev.set_option("shadow_exact", 1)
    dev.set_option("visited_slots", 256)  # force ids into the HBM bitmap
    first = dev.search(Q[:64], 10, 150, stats=True)
    for _ in range(5):
        again = dev.search(Q[:64], 10, 150, stats=True)
        assert np.array_equal(first[1], again[1]) and np.array_equal(first[0], again[0])
        assert np.array_equal(first[2]["n_dist"], again[2]["n_dist"])
    big = dev.search(Q, 10, 150, stats=True)  # and a launch without shadows afterwards sees clean bitmaps
    assert np.array_equal(big[1][:64], first[1]) and np.array_equal(big[2]["n_dist"][:64], first[2]["n_dist"])


def test_views_count_on_their_source(oracle_mod, hipmod):
    X, Q = ds.sift_like(8000, 300)
    ix = oracle_mod.OracleIndex.create("l2", 128, 8000, 16)
    ix.add(X[:4000], 48)
    half_blob = np.array(ix.blob()[: 4000 * ix.node_size], copy=True)
    want_half = ix.search(Q, 10, 50, stats=True)
    ix.add(X[4000:], 48, labels=np.arange(4000, 8000))
    want_full = ix.search(Q, 10, 50, stats=True)
    # a source that is still growing: capacity 8000, the first 4000 nodes live
    src = hipmod.DeviceIndex.alloc(16, 8000, "float32", "l2", 128)
    src.write_nodes(0, half_blob, ix.node_size, ix.data_size)
    src.set_live_nodes(4000)
    view = src.view()
    second = view.view()  # a view of a view hangs off the owner
    with pytest.raises(ValueError, match="live views"):
        src.close()
    for h in (src, view, second):
        got = h.search(Q, 10, 50, stats=True)
        assert np.array_equal(want_half[1], got[1]) and np.array_equal(want_half[0], got[0])
    # the source grows: the views read its live node count at every launch
    src.write_nodes(0, np.asarray(ix.blob()), ix.node_size, ix.data_size)
    src.set_live_nodes(8000)
    for h in (src, view, second):
        got = h.search(Q, 10, 50, stats=True)
        assert np.array_equal(want_full[1], got[1]) and np.array_equal(want_full[0], got[0])
        assert np.array_equal(want_full[2]["n_dist"], got[2]["n_dist"])
    second.close()
    view.close()
    src.close()  # now it goes


def test_replicas_take_over_the_sources_options(oracle_mod, hipmod):
    X, Q = ds.sift_like(8000, 1000)
    ix = oracle_mod.OracleIndex.create("l2", 128, 8000, 16)
    ix.add(X, 48, labels=np.arange(8000) + 500000)  # one thread: node i carries label 500000 + i
    src = _upload(hipmod, ix)
    src.set_option("output_node_ids", 1)
    src.set_option("sorted_beam", 0)
    ids = src.search(Q, 10, 50)
    assert ids[1].max() < 8000
    reps = src.replicate([0, 0])
    got = hipmod.search_multi([src] + reps, Q, 10, 50)
    assert np.array_equal(got[1], ids[1]) and np.array_equal(got[0], ids[0])
    for r in reps:
        r.search(Q[:200], 10, 50)
        assert r.launch_geometry()["kernel"] == "two_heaps"
    # options changed later reach the replicas with the next refresh; until then a mixed batch is refused
    src.set_option("output_node_ids", 0)
    with pytest.raises(ValueError, match="output_node_ids"):
        hipmod.search_multi([src] + reps, Q, 10, 50)
    src.refresh_replicas(reps)
    got = hipmod.search_multi([src] + reps, Q, 10, 50)
    assert got[1].min() >= 500000 and np.array_equal(got[1] - 500000, ids[1])


def test_multi_handle_search_overlaps_its_shards(oracle_mod, hipmod):
    import torch

    X, Q = ds.sift_like(40000, 40000)
    ix = oracle_mod.OracleIndex.create("l2", 128, 40000, 16)
    ix.add(X, 48)
    src = _upload(hipmod, ix)
    rep = src.replicate([0])[0]
    before = torch.cuda.current_device()
    want = src.search(Q, 10, 200)
    hipmod.search_multi([src, rep], Q, 10, 200)  # warm: workspaces, plans
    overlapped = 0
    for _ in range(5):
        got = hipmod.search_multi([src, rep], Q, 10, 200)
        a, b = src.launch_info(), rep.launch_info()
        # each shard (20 000 queries, 5 MB of pageable queries, several ms of kernel) is enqueued before the other completes
        overlapped += 1 if (b["enqueued_ns"] < a["completed_ns"] and a["enqueued_ns"] < b["completed_ns"]) else 0
        assert np.array_equal(got[1], want[1]) and np.array_equal(got[0], want[0])
    # reported, not asserted: whether two host threads overlap depends on the box's scheduler, not on the library's results
    from conftest import SUMMARY_LINES

    SUMMARY_LINES.append("2 shards x 20 000 queries on one GPU: both in flight together in %d of 5 repetitions (reported, not asserted)"
                         % overlapped)
    assert torch.cuda.current_device() == before


def test_gather_ceiling_is_a_plausible_rate(oracle_mod, hipmod):
    import flatnav_amd as flatnav

    X, _ = ds.sift_like(400_000, 10)
    index = flatnav.index.create("l2", 128, len(X), 16)
    index.set_num_threads(min(16, os.cpu_count() or 1))
    index.add(X, 32, device=True)
    import ctypes

    dev = hipmod.DeviceIndex(ctypes.c_void_p(index.device_handle()), owned=False)
    before = index.search(X[:500], 5, 50)
    rate = dev.gather_ceiling()
    print("gather ceiling on a 205 MB table of 512-byte rows: %.0f GB/s" % rate)
    assert 2000 < rate < 16000  # an Infinity-Cache-resident table may exceed the HBM peak
    after = index.search(X[:500], 5, 50)
    assert np.array_equal(before[0], after[0]) and np.array_equal(before[1], after[1])  # the measurement left the index intact
