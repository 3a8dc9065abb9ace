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
