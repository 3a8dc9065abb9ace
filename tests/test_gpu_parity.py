"""GPU parity: the HIP search path (through the C ABI) against the CPU oracle on the same graphs.

Bar: integer-valued data -> ids, distances, per-query counters BIT-EXACT, including
tie-heavy inputs; float data -> distances within rtol 1e-5 (stated below), >= 99.9% of queries with
identical id lists (anything looser indicates a bug), recall identical to 3 decimals."""
import numpy as np
import pytest

from flatnav_amd import datasets as ds

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hipmod():
    from flatnav_amd import hip

    assert hip.device_count() >= 1, "no MI355X visible"
    return hip


_BUILT = {}


def _build(oracle_mod, metric, dt, X, M, efc=100):
    # single-threaded oracle builds are the slow part of this file on a busy host: build each distinct graph once
    import hashlib

    key = (metric, dt, X.shape, M, efc, hashlib.sha1(np.ascontiguousarray(X).view(np.uint8)).hexdigest())
    if key not in _BUILT:
        ix = oracle_mod.OracleIndex.create(metric, X.shape[1], X.shape[0], M, dt)
        ix.add(X, efc)
        _BUILT[key] = ix
    return _BUILT[key]


def _upload(hipmod, ix):
    return hipmod.DeviceIndex.upload(ix.blob(), ix.node_size, ix.data_size, ix.M, ix.cur_nodes, ix.dtype,
                                     ix.metric, ix.dim)


def _assert_exact(o, g):
    od, ol, ost = o
    gd, gl, gst = g
    assert np.array_equal(ost["count"], gst["count"])
    assert np.array_equal(ol, gl), "ids differ in %d queries" % int((ol != gl).any(axis=1).sum())
    assert np.array_equal(od.view(np.uint32), gd.view(np.uint32))
    assert np.array_equal(ost["n_dist"], gst["n_dist"])
    assert np.array_equal(ost["n_hops"], gst["n_hops"])


@pytest.mark.parametrize("ef", [1, 10, 50, 100, 200])
def test_sift_like_float_exact(oracle_mod, hipmod, ef):
    X, Q = ds.sift_like(20000, 1000)
    ix = _build(oracle_mod, "l2", "float32", X, 32)
    dev = _upload(hipmod, ix)
    _assert_exact(ix.search(Q, 10, ef, stats=True), dev.search(Q, 10, ef, stats=True))


@pytest.mark.parametrize("dim,rng_hi", [(128, 256), (32, 16), (16, 4)])
def test_uint8_tie_densities_exact(oracle_mod, hipmod, dim, rng_hi):
    # d=16 / range 4: EVERY query has ties inside its beam -> exercises the libstdc++-exact heaps
    rng = np.random.default_rng(dim)
    X = rng.integers(0, rng_hi, (8000, dim)).astype(np.uint8)
    Q = rng.integers(0, rng_hi, (1000, dim)).astype(np.uint8)
    ix = _build(oracle_mod, "l2", "uint8", X, 16)
    dev = _upload(hipmod, ix)
    for K, ef in ((10, 50), (1, 10), (20, 100)):
        o = ix.search(Q, K, ef, stats=True)
        if rng_hi <= 16 and K > 1:
            assert (np.diff(o[0], axis=1) == 0).any(), "test data should contain ties"
        _assert_exact(o, dev.search(Q, K, ef, stats=True))


@pytest.mark.parametrize("metric,dt", [("l2", "int8"), ("ip", "int8"), ("ip", "uint8"), ("l2", "uint8")])
def test_integer_dtypes_exact(oracle_mod, hipmod, metric, dt):
    rng = np.random.default_rng(7)
    lo, hi = (0, 64) if dt == "uint8" else (-32, 32)
    X = rng.integers(lo, hi, (6000, 100)).astype(dt)
    Q = rng.integers(lo, hi, (500, 100)).astype(dt)
    ix = _build(oracle_mod, metric, dt, X, 16)
    dev = _upload(hipmod, ix)
    _assert_exact(ix.search(Q, 10, 64, stats=True), dev.search(Q, 10, 64, stats=True))


@pytest.mark.parametrize("dim,M", [(100, 32), (37, 16), (7, 8), (768, 32), (500, 16), (260, 16), (2100, 8)])
def test_dimension_classes_integer_valued(oracle_mod, hipmod, dim, M):
    # one case per kernel configuration (row of 16-byte chunks handled by G lanes x CU loads)
    rng = np.random.default_rng(dim)
    hi = 256 if dim <= 256 else 64  # keep sums < 2^24
    n = 4000 if dim <= 768 else 1500
    X = rng.integers(0, hi, (n, dim)).astype(np.float32)
    Q = rng.integers(0, hi, (300, dim)).astype(np.float32)
    for metric in ("l2", "ip"):
        Xm, Qm = (X, Q) if metric == "l2" else (np.minimum(X, 15), np.minimum(Q, 15))
        ix = _build(oracle_mod, metric, "float32", Xm, M, efc=64)
        dev = _upload(hipmod, ix)
        _assert_exact(ix.search(Qm, 10, 50, stats=True), dev.search(Qm, 10, 50, stats=True))


@pytest.mark.parametrize("metric", ["l2", "ip"])
def test_float_data_within_tolerance(oracle_mod, hipmod, metric):
    # Float contract: rtol 1e-5 on distances, >= 99.9% identical id lists, recall equal to 3 decimals.
    X, Q = ds.randn(20000, 1000, 128, seed=3, normalize=(metric == "ip"))
    ix = _build(oracle_mod, metric, "float32", X, 32)
    dev = _upload(hipmod, ix)
    od, ol = ix.search(Q, 10, 100)
    gd, gl = dev.search(Q, 10, 100)
    same = (ol == gl).all(axis=1)
    print("ids identical to the oracle on %.2f%% of %d queries (%s)" % (100 * same.mean(), len(Q), metric))
    assert same.mean() >= 0.999
    # stated float tolerance: rtol 1e-5 (atol 1e-6 for inner-product distances near zero)
    assert np.allclose(od[same], gd[same], rtol=1e-5, atol=1e-6)
    worst = float(np.max(np.abs(od[same] - gd[same]) / np.maximum(np.abs(od[same]), 1e-3)))
    print("max relative distance deviation GPU vs oracle (%s): %.2e" % (metric, worst))
    gt = ds.exact_topk_l2(X, Q, 10) if metric == "l2" else ds.exact_topk_ip(X, Q, 10)
    assert abs(ds.recall_at_k(ol, gt) - ds.recall_at_k(gl, gt)) < 1e-3
    # Same graph searched with the REFERENCE's own compiled AVX-512 / -ffast-math distance kernel (oracle/_ref, when it
    # was built): the GPU must stay within the same tolerance of the reference's float arithmetic.
    if ix.use_reference_distance(True):
        rd, rl = ix.search(Q, 10, 100)
        same_r = (rl == gl).all(axis=1)
        assert same_r.mean() >= 0.999
        assert np.allclose(rd[same_r], gd[same_r], rtol=1e-5, atol=1e-6)
        print("ids identical to the reference-distance search on %.2f%% of queries" % (100 * same_r.mean()))


def test_small_index_and_short_results(oracle_mod, hipmod):
    rng = np.random.default_rng(0)
    X = rng.integers(0, 50, (150, 8)).astype(np.float32)
    ix = _build(oracle_mod, "l2", "float32", X, 8, efc=50)
    dev = _upload(hipmod, ix)
    o = ix.search(X[:40], 200, 300, stats=True)  # K > N: fewer than K results, padded with (+inf, -1)
    g = dev.search(X[:40], 200, 300, stats=True)
    _assert_exact(o, g)
    assert (g[2]["count"] < 200).all() and (g[1][:, -1] == -1).all() and np.isinf(g[0][:, -1]).all()
    for n_init in (1, 7, 100, 1000):
        _assert_exact(ix.search(X[:40], 5, 20, n_init, stats=True), dev.search(X[:40], 5, 20, n_init, stats=True))


def test_spill_paths_stay_exact(oracle_mod, hipmod):
    # Force the visited set into its HBM bitmap and the candidate heap into its HBM spill area.
    X, Q = ds.sift_like(20000, 600)
    ix = _build(oracle_mod, "l2", "float32", X, 32)
    o = ix.search(Q, 10, 100, stats=True)
    dev = _upload(hipmod, ix)
    dev.set_option("visited_slots", 256)  # 16-bit-tag table, 64 buckets: most ids end up in the bitmap
    dev.set_option("sorted_beam", 1)  # merged-beam kernel, exact re-run of the queries with ties
    _assert_exact(o, dev.search(Q, 10, 100, stats=True))
    assert dev.launch_geometry()["kernel"] == "merged_beam_registers"
    dev.set_option("beam_registers", 0)
    _assert_exact(o, dev.search(Q, 10, 100, stats=True))
    assert dev.launch_geometry()["kernel"] == "merged_beam_lds"
    dev.set_option("sorted_beam", 0)  # from here on the two-heap kernel alone: its spill paths are the subject
    _assert_exact(o, dev.search(Q, 10, 100, stats=True))
    assert dev.launch_geometry()["kernel"] == "two_heaps"
    dev.set_option("visited_wide", 1)     # 32-bit open-addressing table, also far too small
    _assert_exact(o, dev.search(Q, 10, 100, stats=True))
    dev.set_option("visited_slots", 0)    # ... and at its default size
    _assert_exact(o, dev.search(Q, 10, 100, stats=True))
    dev.set_option("visited_wide", 0)
    dev.set_option("cand_slots", 8)  # kernel raises it to B+1, far below the ~2.6*B admissions
    _assert_exact(o, dev.search(Q, 10, 100, stats=True))
    # a second search on the same slots must see clean spill bitmaps
    dev.set_option("visited_slots", 256)
    _assert_exact(o, dev.search(Q, 10, 100, stats=True))
    dev.set_option("spill_entries", 1)
    with pytest.raises(RuntimeError):
        dev.search(Q, 10, 100)


@pytest.mark.parametrize("tag_bits,slots", [(0, 256), (32, 384), (32, 256)])
def test_stash_hands_over_to_the_bitmap_exactly(oracle_mod, hipmod, tag_bits, slots):
    # The visited set's three levels on the DEVICE (tests/test_visited_model.py only models them on the CPU): a table of
    # 256-384 slots at ef = 250 (~4500 ids per query) fills within the first hops, the 64-word stash behind it a few hops
    # later, and from then on every id takes the path "both buckets full -> its stash bucket full -> HBM bitmap" while ids
    # that did get a table or stash slot must still be found there.  16-bit tags, three 21-bit tags and two 32-bit tags per
    # bucket; merged-beam kernel in registers and in LDS, and the two-heap kernel: ids, distance bits and the evaluation /
    # hop counters equal the oracle's (a set that forgot or invented a member would change n_dist).
    X, Q = ds.sift_like(20000, 500)
    ix = _build(oracle_mod, "l2", "float32", X, 32)
    o = ix.search(Q, 10, 250, stats=True)
    dev = _upload(hipmod, ix)
    dev.set_option("visited_tag_bits", tag_bits)
    dev.set_option("visited_slots", slots)
    for sorted_beam, registers in ((1, 1), (1, 0), (0, 1)):
        dev.set_option("sorted_beam", sorted_beam)
        dev.set_option("beam_registers", registers)
        for _ in range(2):  # the second launch finds the slots' bitmaps handed back clean
            _assert_exact(o, dev.search(Q, 10, 250, stats=True))
        g = dev.launch_geometry()
        assert g["visited_slots"] == slots and g["kernel"] == ("two_heaps" if not sorted_beam else "merged_beam_registers" if registers else "merged_beam_lds")


@pytest.mark.parametrize("dt", ["float32", "uint8"])
def test_wide_tag_visited_tables_stay_exact(oracle_mod, hipmod, dt):
    # Indexes beyond 2^24 nodes cannot use 16-bit tags at an affordable table size; the kernel then keeps three
    # 21-bit or two 32-bit tags per 64-bit bucket.  Forced here on a small index: slots = 3*2^j -> 21-bit tags,
    # slots = 2^j -> 32-bit tags; roomy, undersized (ids overflow to the HBM bitmap), with and without the HBM
    # overflow list that big indexes use to clear only the bitmap words they touched.
    X, Q = ds.sift_like(20000, 600)
    X, Q = X.astype(dt), Q.astype(dt)
    ix = _build(oracle_mod, "l2", dt, X, 32)
    dev = _upload(hipmod, ix)
    dev.set_option("visited_tag_bits", 32)
    for ef in (50, 200):
        o = ix.search(Q, 10, ef, stats=True)
        for slots in (0, 3072, 4096, 384, 256):
            dev.set_option("visited_slots", slots)
            for ovf in (0, 64, 16384):
                dev.set_option("overflow_list", ovf)
                _assert_exact(o, dev.search(Q, 10, ef, stats=True))
                g = dev.launch_geometry()
                assert slots == 0 or g["visited_slots"] == slots


@pytest.mark.parametrize("case", ["u8_ties", "sift_f32", "randn_ip", "i8_ip"])
def test_sorted_beam_kernel_stays_exact(oracle_mod, hipmod, case):
    # Merged-beam kernel (default): the beam as one sorted array, one merge per link row -- in registers for beams of
    # at most 256 entries (one- and four-chunk forms), beyond that (or with "beam_registers" = 0) in LDS.  A
    # query in which equal keys meet at a decision is searched again by the same wave with the exact two-heap code
    # (candidates heap in LDS or, when LDS is short, in the HBM spill area); ids, distances, counts and the per-query
    # counters must equal the two-heap kernel's and the oracle's bit for bit.
    rng = np.random.default_rng(12)
    if case == "u8_ties":  # every query ties: nearly everything is searched twice
        X = rng.integers(0, 4, (8000, 16)).astype(np.uint8); Q = rng.integers(0, 4, (600, 16)).astype(np.uint8)
        metric, dt, M = "l2", "uint8", 16
    elif case == "sift_f32":  # integer-valued floats: a few per cent
        X, Q = ds.sift_like(20000, 600); metric, dt, M = "l2", "float32", 32
    elif case == "i8_ip":
        X = rng.integers(-20, 20, (6000, 40)).astype(np.int8); Q = rng.integers(-20, 20, (400, 40)).astype(np.int8)
        metric, dt, M = "ip", "int8", 16
    else:  # float data: (almost) never
        X, Q = ds.randn(20000, 600, 96, seed=4, normalize=True); metric, dt, M = "ip", "float32", 32
    ix = _build(oracle_mod, metric, dt, X, M)
    dev = _upload(hipmod, ix)
    for K, ef in ((10, 100), (10, 65), (10, 64), (1, 1), (5, 17), (64, 64), (100, 100), (10, 128), (10, 129), (10, 200),
                  (300, 300), (10, 1000)):
        dev.set_option("sorted_beam", 0)
        want = dev.search(Q, K, ef, stats=True)
        assert dev.replayed_queries()["total"] == 0 and dev.launch_geometry()["kernel"] == "two_heaps"
        if case != "randn_ip":
            _assert_exact(ix.search(Q, K, ef, stats=True), want)
        dev.set_option("sorted_beam", 1)
        auto = "merged_beam_registers" if max(K, ef) <= 256 else "merged_beam_lds"
        forms = [(auto, 2, 1), (auto, 0, 1), (None, 1, 1), ("merged_beam_lds", 2, 0), ("merged_beam_lds", 0, 0)]
        for kernel, cand_lds, regs in forms:
            dev.set_option("beam_registers", regs)
            dev.set_option("sorted_cand_lds", cand_lds)  # 0: the exact re-run keeps its candidates heap in HBM
            got = dev.search(Q, K, ef, stats=True)
            g = dev.launch_geometry()
            assert kernel is None or g["kernel"] == kernel
            assert (g["cand_slots"] == 0) == (cand_lds == 0) or cand_lds == 2
            _assert_exact(want, got)
            r = dev.replayed_queries()
            assert r["total"] == r["eviction_tie"] + r["selection_tie"] + r["result_tie"] + r["nan_inf"] <= len(Q)
            if case == "u8_ties" and ef >= 17:
                assert r["total"] > len(Q) // 2
            if case == "randn_ip" and ef <= 200:
                assert r["total"] <= len(Q) // 20
    dev.set_option("sorted_cand_lds", 2)
    dev.set_option("beam_registers", 1)
    dev.set_option("visited_slots", 256)  # visited ids overflow into the HBM bitmap, in the first pass and the re-run
    _assert_exact(want, dev.search(Q, 10, 1000, stats=True))
    _assert_exact(dev.search(Q, 10, 200, stats=True), (lambda: (dev.set_option("sorted_beam", 0), dev.search(Q, 10, 200, stats=True))[1])())
    dev.set_option("sorted_beam", 1)
    dev.set_option("visited_slots", 0)
    # adaptive default: launches of >= 2048 queries are timed, both kernels get their samples, the faster one stays;
    # where (almost) every query ties that is the two-heap kernel.  Whatever runs, the bytes are the same.
    dev.set_option("sorted_beam", 2)
    Qbig = np.tile(Q, (-(-2048 // len(Q)), 1))  # the tuner looks at launches of at least 2048 queries
    first, kernels = None, []
    for _ in range(7):
        got = dev.search(Qbig, 10, 100, stats=True)
        kernels.append(dev.launch_geometry()["kernel"])
        dev.replayed_queries()  # synchronises: the launch's timing is complete when the next call looks at it
        if first is None:
            first = got
        _assert_exact(first, got)
    assert kernels[0] == "merged_beam_registers" and "two_heaps" in kernels
    if case == "u8_ties":
        assert kernels[-1] == "two_heaps"


@pytest.mark.parametrize("case", ["sift_f32", "u8_ties"])
def test_sorted_beam_tail_goes_straight_to_the_exact_search(oracle_mod, hipmod, case):
    # "sorted_tail_exact_pct": the last queries of a launch of more than one round skip the sorted pass (a query that
    # is searched twice in the last round lengthens the launch).  Which queries take which path must not show anywhere.
    rng = np.random.default_rng(3)
    if case == "u8_ties":
        X = rng.integers(0, 4, (8000, 16)).astype(np.uint8); Q = rng.integers(0, 4, (9000, 16)).astype(np.uint8)
        metric, dt, M = "l2", "uint8", 16
    else:
        X, Q = ds.sift_like(20000, 9000); metric, dt, M = "l2", "float32", 32
    ix = _build(oracle_mod, metric, dt, X, M)
    dev = _upload(hipmod, ix)
    for ef in (52, 100):
        want = ix.search(Q, 10, ef, stats=True)
        dev.set_option("sorted_beam", 1)
        for pct in (0, 37, 100, 1000):
            dev.set_option("sorted_tail_exact_pct", pct)
            got = dev.search(Q, 10, ef, stats=True)
            g = dev.launch_geometry()
            assert g["kernel"] == "merged_beam_registers" and g["grid_blocks"] < len(Q)
            assert g["tail_exact"] == min(len(Q), pct * g["grid_blocks"] // 100)
            _assert_exact(want, got)
        # one round only: there is no tail
        dev.search(Q[:1000], 10, ef)
        assert dev.launch_geometry()["tail_exact"] == 0
        # adaptive default: tails of 50 / 75 / 100 % of a round are three more variants that get measured (three samples
        # each); same bytes whichever runs
        dev.set_option("sorted_tail_exact_pct", -1)
        dev.set_option("sorted_beam", 2)
        seen = set()
        for _ in range(16):
            _assert_exact(want, dev.search(Q, 10, ef, stats=True))
            g = dev.launch_geometry()
            seen.add((g["kernel"] == "two_heaps", g["tail_exact"] * 4 // g["grid_blocks"]))
            dev.replayed_queries()  # synchronises: the launch's timing is complete when the next call looks at it
        assert {(True, 0), (False, 0), (False, 2), (False, 3), (False, 4)} <= seen


def test_infinite_distances_go_to_the_exact_search(oracle_mod, hipmod):
    # Some rows lie so far out that their squared distance overflows float32 to +inf.  The merged-beam kernel does not
    # reason about non-finite keys: a query that could admit one is handed to the exact two-heap search, which treats
    # +inf like the reference does (an ordinary key that is never "< max_dist" once the beam is full).
    rng = np.random.default_rng(21)
    X, Q = ds.sift_like(6000, 300)
    far = rng.choice(len(X), 300, replace=False)
    X = X.copy(); X[far, rng.integers(0, 128, 300)] = 3e19
    ix = _build(oracle_mod, "l2", "float32", X, 16)
    dev = _upload(hipmod, ix)
    for ef in (8, 64, 300):
        want = ix.search(Q, 10, ef, stats=True)
        for mode in (1, 0):
            dev.set_option("sorted_beam", mode)
            _assert_exact(want, dev.search(Q, 10, ef, stats=True))
            if mode == 1 and ef == 8:
                assert dev.replayed_queries()["nan_inf"] > 0  # small beams fill up with whatever comes first


def test_merged_beam_forms_agree_with_the_exact_kernel_on_random_shapes(oracle_mod, hipmod, tmp_path):
    # Randomised sweep of the merged-beam kernel (one-chunk / four-chunk register forms, LDS form) against the exact
    # two-heap kernel on GPU-built graphs: element types, metrics, row widths, link-row widths, beam widths and tie
    # densities drawn at random; every sixth graph is also searched by the oracle.  (Trial 18 of this stream is the
    # case that showed the selection-tie rule needed `pend_cut`: three unexpanded members with one key whose rows hold
    # two neighbours with the key of the farthest member -- whichever row comes first keeps its neighbour.)
    import ctypes
    import flatnav_amd as flatnav

    import os
    rng = np.random.default_rng(int(os.environ.get("FNV_FUZZ_SEED", "2026")))
    rng2 = np.random.default_rng(int(os.environ.get("FNV_FUZZ_SEED", "2026")) + 1)  # later additions: own stream
    trials, oracle_every = int(os.environ.get("FNV_FUZZ_TRIALS", "90")), int(os.environ.get("FNV_FUZZ_ORACLE_EVERY", "6"))
    seen = set()
    for trial in range(trials):
        dt = ["float32", "uint8", "int8"][trial % 3]
        metric = ["l2", "angular"][int(rng.integers(0, 2))]
        dim = int(rng.choice([8, 24, 32, 64, 100, 128, 200, 768]))
        M = int(rng.choice([4, 8, 16, 32, 48, 70]))
        N = int(rng.integers(800, 12000))
        spread = int(rng.choice([2, 4, 16, 120]))  # few distinct values -> ties everywhere
        if dt == "int8":
            X = rng.integers(-spread, spread, (N, dim)).astype(np.int8); Q = rng.integers(-spread, spread, (256, dim)).astype(np.int8)
        elif dt == "uint8":
            X = rng.integers(0, 2 * spread, (N, dim)).astype(np.uint8); Q = rng.integers(0, 2 * spread, (256, dim)).astype(np.uint8)
        else:
            X = rng.integers(0, 2 * spread, (N, dim)).astype(np.float32); Q = rng.integers(0, 2 * spread, (256, dim)).astype(np.float32)
        real = None
        if dt == "float32" and trial % 2 == 1:  # real-valued rows: both kernels share the arithmetic, so still bit for bit
            real = rng2.standard_normal((N + 256, dim)).astype(np.float32)
            X, Q = real[:N], real[N:]
        kw = {} if dt == "float32" else {"index_data_type": getattr(flatnav.data_type.DataType, dt)}
        ix = flatnav.index.create(metric, dim, N, M, **kw)
        ix.set_num_threads(4)
        ix.add(X, 40, device=True)
        dev = hipmod.DeviceIndex(ctypes.c_void_p(ix.device_handle()), owned=False)
        oix = None
        if trial % oracle_every == 0 and real is None:  # (real-valued data: the oracle's sums are ordered differently)
            ix.save(str(tmp_path / "fuzz.bin"))
            oix = oracle_mod.OracleIndex.load(str(tmp_path / "fuzz.bin"), "l2" if metric == "l2" else "ip")
        e1, e2 = int(rng2.integers(2, 40)), int(rng2.integers(40, 200))
        dev.set_option("visited_slots", 256 if trial % 4 == 1 else 0)  # every fourth graph: most ids go to the HBM bitmap
        dev.set_option("visited_tag_bits", [0, 0, 32, 0, 21][trial % 5])  # the wide-tag tables of indexes beyond 2^24 nodes
        dev.set_option("entry_kernel", 1 if trial % 7 == 3 else 0)      # entry points from the batch scan kernel (K0)
        n_init = [100, 100, 1, 7, 100000][trial % 5]                    # Index.h:845-870: ceil(N / step) fixed nodes
        for K, ef in ((1, int(rng.integers(1, 9))), (10, int(rng.integers(10, 65))), (int(rng.integers(1, 80)), int(rng.integers(65, 257))),
                      (10, int(rng.integers(257, 700))), (e1, e1), (e2, e2)):  # K == ef: every beam member is a result
            dev.set_option("sorted_beam", 0)
            want = dev.search(Q, K, ef, n_init, stats=True)
            if oix is not None:
                _assert_exact(oix.search(Q, K, ef, n_init, stats=True, threads=8), want)
            dev.set_option("sorted_beam", 1)
            for regs in (1, 0):
                dev.set_option("beam_registers", regs)
                got = dev.search(Q, K, ef, n_init, stats=True)
                name = dev.launch_geometry()["kernel"]
                # ("two_heaps": indexes too small for the tagged visited table the merged-beam kernel needs)
                assert name in ("merged_beam_registers" if regs and max(K, ef) <= 256 else "merged_beam_lds", "two_heaps")
                seen.add(name)
                try:
                    _assert_exact(want, got)
                    if regs and trial % 3 == 0:  # the answer to a query does not depend on what it is batched with
                        for a, b in ((0, 1), (5, 70)):
                            part = dev.search(Q[a:b], K, ef, n_init, stats=True)
                            _assert_exact((want[0][a:b], want[1][a:b], {k: v[a:b] for k, v in want[2].items()}), part)
                except AssertionError as e:
                    raise AssertionError("trial %d: %s %s d=%d M=%d N=%d spread=%d K=%d ef=%d %s: %s" % (
                        trial, dt, metric, dim, M, N, spread, K, ef, name, e))
        dev.set_option("sorted_beam", 2)
        dev.set_option("beam_registers", 1)
        dev.set_option("visited_slots", 0)
        dev.set_option("visited_tag_bits", 0)
        dev.set_option("entry_kernel", 0)
    assert {"merged_beam_registers", "merged_beam_lds"} <= seen


def test_labels_and_duplicate_links(oracle_mod, hipmod):
    rng = np.random.default_rng(5)
    X = rng.integers(0, 256, (3000, 64)).astype(np.float32)
    labels = rng.permutation(3000).astype(np.int32) * 7 - 5000
    ix = oracle_mod.OracleIndex.create("l2", 64, 3000, 16)
    ix.add(X, 64, labels=labels)
    # plant duplicate ids inside link rows (an .mtx import can produce them, Index.h:219-235);
    # the reference sees the second copy as already visited
    blob = ix.blob()
    rows = blob.reshape(3000, ix.node_size)
    links = rows[:, ix.data_size:ix.data_size + 4 * 16].copy().view(np.uint32)
    links[::3, 5] = links[::3, 2]
    rows[:, ix.data_size:ix.data_size + 4 * 16] = links.view(np.uint8)
    dev = _upload(hipmod, ix)
    Q = X[:500] + 1
    _assert_exact(ix.search(Q, 10, 64, stats=True), dev.search(Q, 10, 64, stats=True))


def test_argument_errors(oracle_mod, hipmod):
    X = np.random.default_rng(1).integers(0, 9, (300, 12)).astype(np.float32)
    ix = _build(oracle_mod, "l2", "float32", X, 8, efc=30)
    dev = _upload(hipmod, ix)
    with pytest.raises(ValueError):
        dev.search(X[:4], 3, 10, num_initializations=0)  # Index.h:847-849
    with pytest.raises(ValueError):
        dev.search(X[:4, :5], 3, 10)
    with pytest.raises(ValueError):
        dev.search(X[:4], 0, 10)
    bad = ix.blob().copy().reshape(300, ix.node_size)
    bad[0, ix.data_size:ix.data_size + 4] = np.array([10 ** 6], dtype=np.uint32).view(np.uint8)
    with pytest.raises(RuntimeError):
        hipmod.DeviceIndex.upload(bad, ix.node_size, ix.data_size, ix.M, 300, "float32", "l2", 12)


@pytest.mark.parametrize("M", [1, 2, 3, 48, 64, 80, 130])
def test_wide_link_rows(oracle_mod, hipmod, M):
    # rows wider than one wavefront are expanded 64 links at a time, still in link order
    rng = np.random.default_rng(M)
    X = rng.integers(0, 256, (3000, 32)).astype(np.float32)
    Q = rng.integers(0, 256, (300, 32)).astype(np.float32)
    ix = _build(oracle_mod, "l2", "float32", X, M, efc=80)
    dev = _upload(hipmod, ix)
    _assert_exact(ix.search(Q, 10, 80, stats=True), dev.search(Q, 10, 80, stats=True))


@pytest.mark.parametrize("ef,K", [(600, 10), (1000, 200), (300, 300)])
def test_large_beams(oracle_mod, hipmod, ef, K):
    # big ef: visited table of tens of KB, heaps beyond the 64- and 256-node mask fast paths
    X, Q = ds.sift_like(20000, 60)
    ix = _build(oracle_mod, "l2", "float32", X, 32)
    dev = _upload(hipmod, ix)
    o = ix.search(Q, K, ef, stats=True)
    _assert_exact(o, dev.search(Q, K, ef, stats=True))
    dev.set_option("sorted_beam", 0)  # the two-heap kernel alone (default: sorted beam first, replay on ties)
    _assert_exact(o, dev.search(Q, K, ef, stats=True))


def test_beam_too_large_for_lds_is_an_error(oracle_mod, hipmod):
    X, Q = ds.sift_like(3000, 4)
    ix = _build(oracle_mod, "l2", "float32", X, 16, efc=64)
    dev = _upload(hipmod, ix)
    with pytest.raises(ValueError):
        dev.search(Q, 10, 20000)


@pytest.mark.parametrize("dim,dt,metric", [(128, "float32", "l2"), (768, "float32", "ip"), (100, "float32", "ip"),
                                           (64, "uint8", "l2")])
def test_entry_scan_kernel_matches_in_kernel_scan(oracle_mod, hipmod, dim, dt, metric):
    # K0 (batched, LDS-staged entry-point selection; several LDS tiles at d=768) vs the in-kernel scan
    rng = np.random.default_rng(dim)
    n = 5000
    X = rng.integers(0, 16 if metric == "ip" else 256, (n, dim)).astype(dt)
    Q = rng.integers(0, 16 if metric == "ip" else 256, (333, dim)).astype(dt)
    ix = _build(oracle_mod, metric, dt, X, 16, efc=64)
    dev = _upload(hipmod, ix)
    for n_init in (100, 7, 1000, 5000):
        o = ix.search(Q, 10, 64, n_init, stats=True)
        dev.set_option("entry_kernel", 1)
        _assert_exact(o, dev.search(Q, 10, 64, n_init, stats=True))
        dev.set_option("entry_kernel", 0)
        _assert_exact(o, dev.search(Q, 10, 64, n_init, stats=True))


@pytest.mark.parametrize("dim,dt", [(1, "float32"), (3, "uint8"), (5, "int8"), (33, "float32")])
def test_tiny_and_ragged_rows(oracle_mod, hipmod, dim, dt):
    # rows shorter than / not a multiple of one 16-byte chunk: zero padding must not change distances
    rng = np.random.default_rng(dim)
    lo, hi = (-20, 20) if dt == "int8" else (0, 40)
    X = rng.integers(lo, hi, (700, dim)).astype(dt)
    Q = rng.integers(lo, hi, (90, dim)).astype(dt)
    for metric in ("l2", "ip"):
        ix = _build(oracle_mod, metric, dt, X, 8, efc=40)
        dev = _upload(hipmod, ix)
        _assert_exact(ix.search(Q, 5, 30, stats=True), dev.search(Q, 5, 30, stats=True))


def test_degenerate_sizes(oracle_mod, hipmod):
    rng = np.random.default_rng(9)
    X = rng.integers(0, 9, (3, 6)).astype(np.float32)
    for n in (1, 2, 3):  # single node: every link is a self-loop
        ix = oracle_mod.OracleIndex.create("l2", 6, 3, 4)
        ix.add(X[:n], 10)
        dev = _upload(hipmod, ix)
        _assert_exact(ix.search(X, 2, 5, stats=True), dev.search(X, 2, 5, stats=True))
        _assert_exact(ix.search(X, 1, 1, 1, stats=True), dev.search(X, 1, 1, 1, stats=True))
    # empty batch: nothing to do, nothing written
    d, l = dev.search(np.zeros((0, 6), dtype=np.float32), 3, 10)
    assert d.shape == (0, 3) and l.shape == (0, 3)
    # one query, host path and device path agree
    import torch

    q = torch.from_numpy(X[:1]).cuda()
    dd = torch.empty((1, 2), dtype=torch.float32, device="cuda")
    dl = torch.empty((1, 2), dtype=torch.int32, device="cuda")
    dev.search_device(q.data_ptr(), 1, 2, 5, 100, dd.data_ptr(), dl.data_ptr())
    torch.cuda.synchronize()
    dev.status()
    hd, hl = dev.search(X[:1], 2, 5)
    assert np.array_equal(dd.cpu().numpy(), hd) and np.array_equal(dl.cpu().numpy(), hl)


def test_many_queries_more_than_slots(oracle_mod, hipmod):
    # 20 000 queries over ~3 000 resident slots: the dispenser hands every query out exactly once
    X, Q = ds.sift_like(5000, 20000)
    ix = _build(oracle_mod, "l2", "float32", X, 16, efc=64)
    dev = _upload(hipmod, ix)
    o = ix.search(Q, 10, 40, threads=8, stats=True)
    _assert_exact(o, dev.search(Q, 10, 40, stats=True))
