"""Round 4 on the GPU, all through the C ABI and against the CPU oracle (bit-exact on integer-valued data):
  * rows of exactly 3 KB keep their query in registers (csrc/distance.hpp `query_in_regs`): every kernel form, every element
    type, a dimension that leaves the last chunk partly empty;
  * large host-buffer batches get their results back as one pinned slab; tail shadows (variant 6) can be pinned;
  * `fnv_index_adopt`: a handle on buffers somebody else owns;
  * options that cannot change the launch plan leave fnv_tune's result alone."""
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


def _assert_exact(o, g, what=""):
    od, ol, ost = o
    gd, gl, gst = g
    assert np.array_equal(ol, gl), "%s: ids differ in %d queries" % (what, int((ol != gl).any(axis=1).sum()))
    assert np.array_equal(od.view(np.uint32), gd.view(np.uint32)), what
    for k in ("count", "n_dist", "n_hops"):
        assert np.array_equal(ost[k], gst[k]), (what, k)


@pytest.mark.parametrize("dt,dim,metric", [("float32", 768, "angular"), ("float32", 768, "l2"), ("float32", 760, "l2"),
                                           ("uint8", 3072, "l2"), ("int8", 3060, "angular")])
def test_three_kilobyte_rows_keep_the_query_in_registers(oracle_mod, hipmod, dt, dim, metric):
    # 192-chunk rows: G = 64 lanes x CU = 3 chunks, the query is 12 registers per lane and takes no LDS.  Integer-valued
    # data, so every summation order gives the same bits: ids, distance bits and the per-query counters equal the oracle's
    # in the two-heap kernel, the merged-beam kernel with 1 / 2 / 4 register chunks and in LDS, with the batched entry scan
    # (K0 fills the registers from its LDS tile) and in the device builder's wiring kernels.
    rng = np.random.default_rng(dim)
    N, NQ = 3000, 200
    hi = {"float32": 12, "uint8": 6, "int8": 4}[dt]
    lo = -3 if dt == "int8" else 0
    X = rng.integers(lo, hi, (N, dim)).astype(dt)
    Q = rng.integers(lo, hi, (NQ, dim)).astype(dt)
    ix = oracle_mod.OracleIndex.create(metric, dim, N, 16, dt)
    ix.add(X, 48)
    dev = _upload(hipmod, ix)
    assert dev.row_bytes == 3072
    lds = {}
    for ef, K in ((20, 5), (100, 10), (200, 10), (300, 20)):
        want = ix.search(Q, K, ef, stats=True)
        for name, opts in (("two_heaps", dict(sorted_beam=0)), ("merged_registers", dict(sorted_beam=1, beam_registers=1)),
                           ("merged_lds", dict(sorted_beam=1, beam_registers=0)), ("entry_kernel", dict(sorted_beam=1, entry_kernel=1))):
            for k, v in {**dict(sorted_beam=2, beam_registers=1, entry_kernel=0), **opts}.items():
                dev.set_option(k, v)
            _assert_exact(want, dev.search(Q, K, ef, stats=True), "%s ef=%d" % (name, ef))
            lds[(name, ef)] = dev.launch_geometry()["lds_bytes"]
    # no LDS for the query: a slot is no bigger than that of a 128-d index of the same size (whose 512-byte query IS in LDS)
    if dt == "float32" and dim == 768:
        Xs = rng.integers(0, 12, (N, 128)).astype(np.float32)
        ixs = oracle_mod.OracleIndex.create(metric, 128, N, 16, dt)
        ixs.add(Xs, 48)
        devs = _upload(hipmod, ixs)
        devs.set_option("sorted_beam", 1)
        devs.search(Xs[:64], 10, 100)
        assert lds[("merged_registers", 100)] <= devs.launch_geometry()["lds_bytes"], (lds, devs.launch_geometry())
    # the device builder's kernels read their staged vectors into the same registers: sequential insertion = the oracle's graph
    import flatnav_amd as flatnav

    index = flatnav.index.create(metric, dim, 600, 16, getattr(flatnav.data_type.DataType, dt))
    index.add(X[:600], 48, device=True, device_max_batch=1, device_bootstrap=40)  # one node per device batch = Index::add
    small = oracle_mod.OracleIndex.create(metric, dim, 600, 16, dt)
    small.add(X[:600], 48)
    want_g = np.asarray(small.blob())[: 600 * small.node_size].reshape(600, small.node_size)
    got_g = np.asarray(index._raw_blob())[: 600 * small.node_size].reshape(600, small.node_size)
    bad = np.flatnonzero((want_g != got_g).any(axis=1))
    assert bad.size == 0, "first differing node %d of %d differing" % (bad[0], bad.size)


@pytest.mark.parametrize("dt", ["float32", "uint8"])
def test_large_host_batches_come_back_through_the_pinned_result_slab(oracle_mod, hipmod, dt):
    # Host-buffer batches above the 1 MB pinned buffer: queries in as one pageable copy, the five result arrays back as ONE
    # slab into pinned memory, scattered by the CPU (round 4).  Whatever the batch size, kernel variant (tuned: an exact
    # tail; pinned ones, tail shadows included) or beam width, ids, distances, counts and counters are the oracle's;
    # consecutive calls with DIFFERENT queries never see each other's data; null output arrays are left alone.
    X, Q = ds.sift_like(20000, 24000)
    X, Q = X.astype(dt), Q.astype(dt)
    Qa, Qb = Q[:12000], Q[12000:]
    ix = oracle_mod.OracleIndex.create("l2", 128, 20000, 16, dt)
    ix.add(X, 48)
    dev = _upload(hipmod, ix)
    K, ef = 10, 64
    want_a = ix.search(Qa, K, ef, stats=True, threads=8)
    want_b = ix.search(Qb, K, ef, stats=True, threads=8)

    def cut(w, n):
        return tuple(x[:n] if not isinstance(x, dict) else {k: v[:n] for k, v in x.items()} for x in w)

    for nq in (12000, 3585, 2049, 7681, 1024 if dt == "uint8" else 2100):
        _assert_exact(cut(want_a, nq), dev.search(Qa[:nq], K, ef, stats=True), "nq=%d" % nq)
        _assert_exact(cut(want_b, nq), dev.search(Qb[:nq], K, ef, stats=True), "other queries, nq=%d" % nq)
    d_only, l_only = dev.search(Qa[:5000], K, ef)  # no count / counter arrays
    assert np.array_equal(l_only, want_a[1][:5000]) and np.array_equal(d_only.view(np.uint32), want_a[0][:5000].view(np.uint32))
    dev.tune(Qa[:10000], K, ef)  # integer-valued data: a tail variant wins
    _assert_exact(want_a, dev.search(Qa, K, ef, stats=True), "tuned")
    for variant in (0, 1, 3, 6):  # 6: tail shadows (pinnable only)
        dev.set_option("sorted_variant", variant)
        _assert_exact(want_b, dev.search(Qb, K, ef, stats=True), "variant %d" % variant)
        assert dev.launch_info()["variant_id"] == variant
    dev.set_option("sorted_variant", -1)
    view = dev.view()
    _assert_exact(want_a, view.search(Qa, K, ef, stats=True), "view")
    view.close()
    # wide beams (LDS form of the merged-beam kernel, spills of the visited set), then the two-heap kernel
    dev.set_option("visited_slots", 512)
    w2 = ix.search(Qa[:3000], 5, 300, stats=True, threads=8)
    _assert_exact(w2, dev.search(Qa[:3000], 5, 300, stats=True), "ef=300")
    dev.set_option("sorted_beam", 0)
    _assert_exact(w2, dev.search(Qa[:3000], 5, 300, stats=True), "ef=300, two heaps")


def test_concurrent_host_callers_share_a_handle(oracle_mod, hipmod):
    # Two (and more) threads in fnv_search_batch on ONE handle: the second caller runs on the handle's hidden second lane
    # (own stream, workspace, staging; same HBM buffers), a third one waits.  Every caller gets the oracle's bytes for ITS
    # queries, fnv_tune and a device insertion keep both lanes out while they run, and the lane takes over options set later.
    import threading

    X, Q = ds.sift_like(20000, 24000)
    ix = oracle_mod.OracleIndex.create("l2", 128, 20000, 16)
    ix.add(X, 48)
    dev = _upload(hipmod, ix)
    K, ef = 10, 64
    parts = [Q[i * 6000:(i + 1) * 6000] for i in range(4)]
    want = [ix.search(p, K, ef, stats=True, threads=8) for p in parts]
    errors = []

    def caller(i, rounds):
        try:
            for _ in range(rounds):
                _assert_exact(want[i], dev.search(parts[i], K, ef, stats=True), "caller %d" % i)
        except Exception as exc:  # noqa: BLE001
            errors.append((i, repr(exc)))

    for n_threads in (2, 4):
        th = [threading.Thread(target=caller, args=(i, 4)) for i in range(n_threads)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errors, errors
    # tuning from one thread while others search: exclusive, and afterwards both lanes run the tuned choice
    th = [threading.Thread(target=caller, args=(i, 6)) for i in (1, 2)]
    for t in th:
        t.start()
    dev.tune(parts[0], K, ef)
    for t in th:
        t.join()
    assert not errors, errors
    # single queries from eight threads: up to eight in flight (lanes 2 ... 7 serve batches of <= 1024 queries only)
    singles = ix.search(Q[:64], K, ef, stats=True)

    def single_caller(i):
        try:
            for j in range(i, 64, 8):
                got = dev.search(Q[j:j + 1], K, ef, stats=True)
                assert np.array_equal(got[1][0], singles[1][j]) and np.array_equal(got[0][0].view(np.uint32), singles[0][j].view(np.uint32))
        except Exception as exc:  # noqa: BLE001
            errors.append((i, repr(exc)))

    th = [threading.Thread(target=single_caller, args=(i,)) for i in range(8)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    dev.set_option("sorted_beam", 0)  # an option set later reaches the second lane as well
    th = [threading.Thread(target=caller, args=(i, 3)) for i in (0, 3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    assert dev.launch_geometry()["kernel"] == "two_heaps"
    dev.close()  # frees the hidden lane first, then the handle


def test_adopted_buffers_answer_like_their_owner(oracle_mod, hipmod):
    # fnv_index_adopt: a second, independent handle (own workspace, stream, options) on buffers the first one owns
    X, Q = ds.sift_like(8000, 700)
    ix = oracle_mod.OracleIndex.create("l2", 128, 8000, 16)
    ix.add(X, 48)
    want = ix.search(Q, 10, 80, stats=True)
    owner = _upload(hipmod, ix)
    guest = hipmod.DeviceIndex.adopt(owner.device_buffers(), owner.M, owner.n_nodes, "float32", "l2", 128, keep_alive=owner)
    assert guest.device_buffers() == owner.device_buffers() and guest.row_bytes == owner.row_bytes
    guest.set_option("sorted_beam", 0)  # its options are its own
    _assert_exact(want, guest.search(Q, 10, 80, stats=True), "guest")
    assert guest.launch_geometry()["kernel"] == "two_heaps"
    _assert_exact(want, owner.search(Q, 10, 80, stats=True), "owner")
    assert owner.launch_geometry()["kernel"] != "two_heaps"
    guest.close()  # never frees what it does not own
    _assert_exact(want, owner.search(Q, 10, 80, stats=True), "owner after the guest left")
    with pytest.raises(ValueError):
        hipmod.DeviceIndex.adopt([(owner.device_buffers()[0][0] + 4, 0)] + owner.device_buffers()[1:], owner.M, owner.n_nodes,
                                 "float32", "l2", 128)  # a misaligned vector table


def test_unrelated_options_leave_the_tuning_alone(oracle_mod, hipmod):
    # Index.h::addBatchDevice flips output_node_ids around every device build; that (and shadow_exact,
    # tune_layout) must not throw fnv_tune's measurements away -- options that change the plan still do.  (Device-pointer
    # entry point: that is where the adaptive choice lives; the host pipeline never runs an exploratory launch.)
    import torch

    X, Q = ds.sift_like(20000, 4096)
    ix = oracle_mod.OracleIndex.create("l2", 128, 20000, 16)
    ix.add(X, 48)
    dev = _upload(hipmod, ix)
    dq = torch.from_numpy(Q).cuda()
    od = torch.empty((4096, 10), dtype=torch.float32, device="cuda")
    ol = torch.empty((4096, 10), dtype=torch.int32, device="cuda")

    def launch():
        dev.search_device(dq.data_ptr(), 4096, 10, 64, 100, od.data_ptr(), ol.data_ptr())
        torch.cuda.synchronize()
        return dev.launch_info()

    dev.tune(int(dq.data_ptr()), 10, 64, nq=4096)
    settled = launch()
    assert not settled["exploratory"]
    for name in ("output_node_ids", "shadow_exact", "tune_layout"):
        dev.set_option(name, 0)
        dev.set_option(name, 1)
    dev.set_option("output_node_ids", 0)
    again = launch()
    assert not again["exploratory"] and again["variant_id"] == settled["variant_id"]
    dev.set_option("visited_factor", 20)  # changes the table size rule: measurements are void
    assert launch()["exploratory"]
