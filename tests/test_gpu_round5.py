"""Round 5 on the GPU, all through the C ABI and against the CPU oracle (bit-exact on integer-valued data):
  * `fnv_set_option` from one thread while four threads search the same handle (the option block, the
    tuner and the layouts are read and written under the handle's mutex; lanes copy them under it);
  * hidden lanes are admitted by the HBM their workspace needs against a budget, idle lanes give their room back
    (ADVICE r4, medium), lanes never run exploratory launches;
  * the launch record a lane leaves on the handle is one launch's record."""
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from flatnav_amd import datasets as ds

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hipmod():
    from flatnav_amd import hip

    assert hip.device_count() >= 1, "no MI355X visible"
    return hip


def _upload(hipmod, ix):
    return hipmod.DeviceIndex.upload(ix.blob(), ix.node_size, ix.data_size, ix.M, ix.cur_nodes, ix.dtype, ix.metric, ix.dim)


def _assert_exact(o, g, what=""):
    od, ol, ost = o
    gd, gl, gst = g
    assert np.array_equal(ol, gl), "%s: ids differ in %d queries" % (what, int((ol != gl).any(axis=1).sum()))
    assert np.array_equal(od.view(np.uint32), gd.view(np.uint32)), what
    for k in ("count", "n_dist", "n_hops"):
        assert np.array_equal(ost[k], gst[k]), (what, k)


def test_options_flip_while_four_threads_search_one_handle(oracle_mod, hipmod):
    # Every option below changes the launch (kernel, LDS layout, tail, table size) but never the result: whatever mix of
    # "before" and "after" the four searching threads see, each call returns the oracle's bytes -- and nothing crashes
    # (round 4: fnv_set_option cleared std::map members that a concurrent caller's lane was copying).
    X, Q = ds.sift_like(20000, 12000)
    ix = oracle_mod.OracleIndex.create("l2", 128, 20000, 16)
    ix.add(X, 48)
    dev = _upload(hipmod, ix)
    K, ef = 10, 64
    parts = [Q[i * 3000:(i + 1) * 3000] for i in range(4)]
    want = [ix.search(p, K, ef, stats=True, threads=8) for p in parts]
    dev.tune(parts[0], K, ef)
    errors, stop = [], threading.Event()

    def caller(i):
        try:
            n = 0
            while not stop.is_set() or n < 3:
                _assert_exact(want[i], dev.search(parts[i], K, ef, stats=True), "caller %d, call %d" % (i, n))
                n += 1
        except Exception as exc:  # noqa: BLE001
            errors.append((i, repr(exc)))

    th = [threading.Thread(target=caller, args=(i,)) for i in range(4)]
    for t in th:
        t.start()
    flips = [("sorted_beam", (0, 1, 2)), ("beam_registers", (0, 1)), ("sorted_variant", (3, 0, 4, -1)), ("visited_slots", (1024, 3072, 0)),
             ("sorted_cand_lds", (0, 1, 2)), ("shadow_exact", (0, 1)), ("blocks_per_cu", (6, 0)), ("output_node_ids", (0,))]
    try:
        for rnd in range(12):
            for name, values in flips:
                dev.set_option(name, values[rnd % len(values)])
                dev.launch_info(), dev.launch_geometry()  # the getters may race as well: each is one launch's record
        for name, values in flips:
            dev.set_option(name, values[-1])
    finally:
        stop.set()
        for t in th:
            t.join()
    assert not errors, errors
    _assert_exact(want[0], dev.search(parts[0], K, ef, stats=True), "after the flips")
    dev.close()


def test_lanes_are_admitted_by_their_workspace_and_give_it_back(oracle_mod, hipmod):
    # A budget of 8 MB for all hidden lanes together: a 20 000-node index needs 2.5 KB of bitmap + 128 KB of spill area per
    # slot, so a 16-query batch (32 slots with its shadows: 4.2 MB) fits ONE lane at a time -- the second concurrent caller
    # must wait for the handle or for a lane whose idle neighbour gave its workspace back -- and a 3000-query batch (a full
    # grid: 500+ MB) fits none: those callers take turns on the handle.  Results are the oracle's either way.
    env = dict(os.environ, FLATNAV_LANE_BUDGET_MB="8")
    code = r"""
import sys, threading, numpy as np
sys.path.insert(0, %r)
from flatnav_amd import datasets as ds, hip
from oracle import oracle as orc
orc.build()
X, Q = ds.sift_like(20000, 6200)
ix = orc.OracleIndex.create("l2", 128, 20000, 16); ix.add(X, 48)
dev = hip.DeviceIndex.upload(ix.blob(), ix.node_size, ix.data_size, ix.M, ix.cur_nodes, ix.dtype, ix.metric, ix.dim)
K, ef = 10, 64
big = [Q[i * 3000:(i + 1) * 3000] for i in range(2)]
small = [Q[6000 + i * 16:6000 + (i + 1) * 16] for i in range(8)]
want_big = [ix.search(p, K, ef, threads=8) for p in big]
want_small = [ix.search(p, K, ef) for p in small]
errors = []
def run(parts, want, i, rounds):
    try:
        for _ in range(rounds):
            d, l = dev.search(parts[i], K, ef)
            assert np.array_equal(l, want[i][1]) and np.array_equal(d.view(np.uint32), want[i][0].view(np.uint32)), i
    except Exception as exc:
        errors.append(repr(exc))
for parts, want, n, rounds in ((big, want_big, 2, 4), (small, want_small, 8, 30), (big, want_big, 2, 2)):
    th = [threading.Thread(target=run, args=(parts, want, i, rounds)) for i in range(n)]
    [t.start() for t in th]; [t.join() for t in th]
    assert not errors, errors
    used = hip.lane_workspaces(dev)
    assert sum(used[1:]) <= 8 << 20, used  # the hidden lanes together never exceed the budget
    print("lanes hold", used)
dev.close()
print("OK")
""" % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_lanes_never_explore(oracle_mod, hipmod):
    # an UNTUNED handle whose callers land on lanes: the lanes run the merged-beam kernel (or whatever the owner has measured
    # so far) and never mark a launch exploratory -- the owner's own launches are the only samples of the adaptive choice
    X, Q = ds.sift_like(20000, 12000)
    ix = oracle_mod.OracleIndex.create("l2", 128, 20000, 16)
    ix.add(X, 48)
    dev = _upload(hipmod, ix)
    K, ef = 10, 48
    parts = [Q[i * 3000:(i + 1) * 3000] for i in range(4)]
    want = [ix.search(p, K, ef, stats=True, threads=8) for p in parts]
    errors, seen = [], []

    def caller(i):
        try:
            for _ in range(6):
                _assert_exact(want[i], dev.search(parts[i], K, ef, stats=True), "caller %d" % i)
        except Exception as exc:  # noqa: BLE001
            errors.append((i, repr(exc)))

    th = [threading.Thread(target=caller, args=(i,)) for i in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    explored = hipmod.lane_exploratory_launches(dev)
    assert explored[0] >= 0 and all(e == 0 for e in explored[1:]), explored
    dev.close()


@pytest.mark.parametrize("case", ["u8_dense_ties", "u8_ties", "sift_f32", "i8_ip"])
def test_tied_queries_are_resumed_from_their_log(oracle_mod, hipmod, case):
    # The mid-flight hand-over (csrc/kernels.hpp `replay_log`): a query in which equal keys meet at a decision is not
    # searched again from scratch -- the reference's two heaps are replayed from the log the merged-beam pass wrote and
    # the exact search continues from there (from the hop where the reference would have expanded another node of equal
    # distance first, with the visited set rebuilt for the hops before it).  ids, distance bits, counts and the per-query
    # counters equal the oracle's in every beam form and heap home, with the log on, off and overflowing.
    rng = np.random.default_rng(5)
    if case == "u8_dense_ties":  # every query ties, early and often: most hand-overs rewind
        X = rng.integers(0, 4, (8000, 16)).astype(np.uint8); Q = rng.integers(0, 4, (500, 16)).astype(np.uint8)
        metric, dt, M = "l2", "uint8", 16
    elif case == "u8_ties":  # about half the queries tie
        X = rng.integers(0, 16, (8000, 32)).astype(np.uint8); Q = rng.integers(0, 16, (500, 32)).astype(np.uint8)
        metric, dt, M = "l2", "uint8", 16
    elif case == "sift_f32":  # integer-valued floats: a few per cent, late in the search -- the bench's case
        X, Q = ds.sift_like(20000, 1500); metric, dt, M = "l2", "float32", 32
    else:
        X = rng.integers(-20, 20, (6000, 40)).astype(np.int8); Q = rng.integers(-20, 20, (400, 40)).astype(np.int8)
        metric, dt, M = "ip", "int8", 16
    ix = oracle_mod.OracleIndex.create(metric, X.shape[1], len(X), M, dt)
    ix.add(X, 48)
    dev = _upload(hipmod, ix)
    dev.set_option("sorted_beam", 1)
    dev.set_option("sorted_variant", 1)  # the merged-beam kernel for every query, no exact tail
    dev.set_option("shadow_exact", 0)    # (launches this small would otherwise start an exact shadow next to every query, which answers the tied ones)
    total_resumed = 0
    for K, ef in ((10, 52), (10, 100), (1, 8), (64, 64), (10, 129), (10, 200), (10, 300), (30, 700)):
        want = ix.search(Q, K, ef, stats=True, threads=8)
        for regs, cand_lds in ((1, 2), (1, 0), (0, 2), (0, 0)):
            dev.set_option("beam_registers", regs)
            dev.set_option("sorted_cand_lds", cand_lds)
            what = "%s K=%d ef=%d regs=%d heap home %d" % (case, K, ef, regs, cand_lds)
            dev.set_option("tie_replay", 1)
            dev.set_option("tie_log_entries", 0)
            _assert_exact(want, dev.search(Q, K, ef, stats=True), what)
            r, h = dev.replayed_queries(), dev.handover_stats()
            assert h["resumed"] + h["from_scratch"] == r["total"] and r["nan_inf"] == 0, (what, r, h)
            # (every hand-over resumes from its log unless the log overflowed: where many distances are equal a tied query
            #  expands hundreds to thousands of nodes more than its beam width suggests -- the automatic log size is made for
            #  data like the bench's, where it never overflows; the tie-dense cases resume with the largest log, below)
            assert h["from_scratch"] == 0 or case != "sift_f32", (what, r, h)
            assert h["hops_from_log"] <= h["hops_of_resumed"], (what, h)
            total_resumed += h["resumed"]
            if case == "sift_f32" and h["resumed"]:  # ties are rare and late: nearly every hop comes from the log
                assert h["hops_from_log"] >= 0.25 * h["hops_of_resumed"], (what, h)
            dev.set_option("tie_replay", 0)  # as in rounds 2-4: searched again from scratch
            _assert_exact(want, dev.search(Q, K, ef, stats=True), what + ", no replay")
            h0 = dev.handover_stats()
            assert h0["resumed"] == 0 and h0["from_scratch"] == r["total"], (what, h0)
            dev.set_option("tie_replay", 1)
            dev.set_option("tie_log_entries", 16384)  # the largest log
            _assert_exact(want, dev.search(Q, K, ef, stats=True), what + ", largest log")
            total_resumed += dev.handover_stats()["resumed"]
            dev.set_option("tie_log_entries", 130)  # two hops' worth: the log ends early, those queries start again
            _assert_exact(want, dev.search(Q, K, ef, stats=True), what + ", tiny log")
            h1 = dev.handover_stats()
            assert h1["resumed"] + h1["from_scratch"] == r["total"], (what, h1)
    assert total_resumed > (1000 if case.startswith("u8") else 20), total_resumed
    print("%s: %d hand-overs resumed from their logs" % (case, total_resumed))
    # the visited set overflows into the stash and the HBM bitmap during the merged-beam pass: a rewind has to give the
    # bitmap back clean before it rebuilds the set (16-, 21- and 32-bit tags)
    dev.set_option("tie_log_entries", 0)
    dev.set_option("beam_registers", 1)
    dev.set_option("sorted_cand_lds", 2)
    for bits in (0, 21, 32):
        dev.set_option("visited_tag_bits", bits)
        for slots in (256, 384):
            dev.set_option("visited_slots", slots)
            for K, ef in ((10, 250), (10, 60)):
                want = ix.search(Q, K, ef, stats=True, threads=8)
                _assert_exact(want, dev.search(Q, K, ef, stats=True), "%s tags %d slots %d ef %d" % (case, bits, slots, ef))
                _assert_exact(want, dev.search(Q, K, ef, stats=True), "%s tags %d slots %d ef %d, again (bitmaps clean?)" % (case, bits, slots, ef))
    dev.close()


def test_handover_in_a_full_launch_and_in_small_ones(oracle_mod, hipmod):
    # the bench shape in small: more queries than slots (rounds, an exact tail of every length, the adaptive choice after
    # fnv_tune), and launches of 1 ... 300 queries (shadow mode on and off) -- every variant returns the oracle's bytes with
    # the hand-over log on
    X, Q = ds.sift_like(30000, 9000)
    Xu, Qu = X.astype(np.uint8), Q.astype(np.uint8)
    for dt, XX, QQ in (("float32", X, Q), ("uint8", Xu, Qu)):
        ix = oracle_mod.OracleIndex.create("l2", 128, len(XX), 32, dt)
        ix.add(XX, 64, threads=1)
        dev = _upload(hipmod, ix)
        dev.set_option("blocks_per_cu", 4)  # 1024 slots: 9000 queries are nine rounds
        K, ef = 10, 52
        want = ix.search(QQ, K, ef, stats=True, threads=8)
        for variant in (1, 2, 3, 4, 5, 6, 0, -1):
            dev.set_option("sorted_variant", variant)
            if variant == -1:
                dev.tune(QQ[:4096], K, ef)
            _assert_exact(want, dev.search(QQ, K, ef, stats=True), "%s variant %d" % (dt, variant))
            if variant == 1:
                r, h = dev.replayed_queries(), dev.handover_stats()
                assert h["resumed"] == r["total"] > 0 and h["from_scratch"] == 0, (r, h)
        dev.set_option("blocks_per_cu", 0)
        for shadow in (1, 0):
            dev.set_option("shadow_exact", shadow)
            for nq in (1, 7, 64, 300):
                got = dev.search(QQ[:nq], K, ef, stats=True)
                _assert_exact(tuple(a[:nq] if not isinstance(a, dict) else {k: v[:nq] for k, v in a.items()} for a in want), got,
                              "%s %d queries, shadow %d" % (dt, nq, shadow))
        dev.close()


@pytest.mark.parametrize("M", [63, 64, 100])
def test_handover_log_at_the_row_width_limit(oracle_mod, hipmod, M):
    # rows of up to 63 links are logged (a hop is at most 64 records, lane 63 writes the header); wider rows write no log and a
    # tied query is searched again from scratch, as in rounds 2-4 -- the oracle's bytes either way, on data where most queries tie
    rng = np.random.default_rng(M)
    X = rng.integers(0, 12, (3000, 24)).astype(np.uint8)
    Q = rng.integers(0, 12, (400, 24)).astype(np.uint8)
    ix = oracle_mod.OracleIndex.create("l2", 24, len(X), M, "uint8")
    ix.add(X, 2 * M)
    dev = _upload(hipmod, ix)
    dev.set_option("sorted_beam", 1)
    dev.set_option("sorted_variant", 1)
    dev.set_option("shadow_exact", 0)
    dev.set_option("tie_log_entries", 16384)
    for K, ef in ((10, 40), (10, 100), (5, 300)):
        want = ix.search(Q, K, ef, stats=True, threads=8)
        _assert_exact(want, dev.search(Q, K, ef, stats=True), "M=%d ef=%d" % (M, ef))
        r, h = dev.replayed_queries(), dev.handover_stats()
        assert r["total"] > 20 and h["resumed"] + h["from_scratch"] == r["total"], (M, r, h)
        if M <= 63:
            assert h["resumed"] > 0.5 * r["total"], (M, r, h)
        else:
            assert h["resumed"] == 0, (M, r, h)
    dev.close()


def test_small_launches_keep_the_visited_set_as_an_lds_bitmap(oracle_mod, hipmod):
    # Round 5: a launch that fills at most a quarter of the slots keeps its visited set as a plain bitmap of all node ids in LDS
    # when that fits (csrc/visited.hpp visited_insert_direct) -- one LDS round trip per link row, nothing overflows -- for the
    # merged-beam pass, its exact shadows, hand-overs (a rewind rebuilds the bitmap) and straight exact searches alike.  The
    # oracle's bytes with it and without it ("visited_direct" = 0), on tie-free, tie-heavy and every-query-ties data.
    rng = np.random.default_rng(11)
    X, Q = ds.sift_like(20000, 600)
    cases = [("sift_f32", X, Q, "l2", "float32", 32),
             ("u8_ties", rng.integers(0, 16, (8000, 32)).astype(np.uint8), rng.integers(0, 16, (600, 32)).astype(np.uint8), "l2", "uint8", 16),
             ("u8_dense_ties", rng.integers(0, 4, (8000, 16)).astype(np.uint8), rng.integers(0, 4, (600, 16)).astype(np.uint8), "l2", "uint8", 16),
             ("i8_ip", rng.integers(-20, 20, (6000, 40)).astype(np.int8), rng.integers(-20, 20, (600, 40)).astype(np.int8), "ip", "int8", 16)]
    for name, XX, QQ, metric, dt, M in cases:
        ix = oracle_mod.OracleIndex.create(metric, XX.shape[1], len(XX), M, dt)
        ix.add(XX, 48)
        dev = _upload(hipmod, ix)
        for K, ef in ((10, 52), (10, 200), (1, 8), (64, 64), (30, 700)):
            want = ix.search(QQ, K, ef, stats=True, threads=8)
            for shadow in (1, 0):
                dev.set_option("shadow_exact", shadow)
                for nq in (1, 7, 64, 300, 600):
                    w = tuple(a[:nq] if not isinstance(a, dict) else {k: v[:nq] for k, v in a.items()} for a in want)
                    lds = {}
                    for direct in (1, 0):
                        dev.set_option("visited_direct", direct)
                        _assert_exact(w, dev.search(QQ[:nq], K, ef, stats=True), "%s K=%d ef=%d %d queries shadow %d direct %d" % (name, K, ef, nq, shadow, direct))
                        g = dev.launch_geometry()
                        lds[direct] = (g["lds_bytes"], g["visited_slots"])
                    # the bitmap is one bit per node id (rounded up to 16 bytes), whatever the beam width
                    bits = (len(XX) + 127) // 128 * 128
                    assert lds[0][1] != bits and (lds[1][1] == bits or nq > 64), (name, nq, lds)
        # a pinned table shape wins over the bitmap (the tests that force the table into stash and HBM bitmap keep doing so)
        dev.set_option("visited_direct", 1)
        dev.set_option("visited_slots", 256)
        want = ix.search(QQ[:64], 10, 100, stats=True, threads=8)
        _assert_exact(want, dev.search(QQ[:64], 10, 100, stats=True), name + " pinned table")
        assert dev.launch_geometry()["visited_slots"] == 256
        dev.close()
