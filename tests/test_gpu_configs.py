"""The BASELINE.json configurations beyond C2 under -m gpu, at sizes the CPU oracle handles in seconds, and -- property
tests at the bottom -- at their FULL sizes (10M x 768, 50M x 128) whenever the box can hold them (a GPU with >= 200 GB
and >= 160 GB of host memory: the MI355X boxes; FNV_FULLSIZE=0 / 1 overrides the probe):
  C3  768-d float inner product on unit vectors (randn as worded, and the recall-qualified low-rank S3), M=32, ef=200
  C4  100-d unit vectors (GloVe stand-in), ef in {50, 100, 200, 400}: rows that are not a whole number of lane spans
  C5  128-d randn L2
Float contract: distances within rtol 1e-5 (atol 1e-6 near zero), >= 99.9 % of queries with identical id
lists -- against the oracle and against the oracle driving the reference's own compiled AVX-512 distance
kernel; the measured fraction is printed."""
import os

import numpy as np
import pytest

from flatnav_amd import datasets as ds

pytestmark = pytest.mark.gpu
ID_BAR = 0.999
# GPU vs the same graph searched with the reference's own compiled distance kernel (oracle/_ref: AVX-512 lanes, -ffast-math --
# a third summation order, neither the oracle's nor the GPU's).  Round 4: every oracle graph of the GPU suite is built by ONE
# thread, so inputs -- and the fractions printed below -- are the same run after run.  On 768-d random unit vectors every
# distance lies within a few per cent of 1 and last-bit differences between summation orders reorder near ties in 1-3 of 1000
# queries: that data set keeps its own bar (REF_ORDER_BAR_RANDN_768), every other shape meets the survey's 99.9 %.  The parity
# claim proper is ID_BAR (GPU vs oracle); these bound the sensitivity to the summation order.
REF_ORDER_BAR = 0.999
REF_ORDER_BAR_RANDN_768 = 0.995
# what the fixture-scoped summary prints at the end of the run (conftest.pytest_terminal_summary): sizes and id fractions
from conftest import SUMMARY_LINES  # noqa: E402


def _fullsize() -> bool:
    """Run the property tests at 10M x 768 / 50M x 128?  FNV_FULLSIZE=0/1 decides; unset: yes when the GPU has >= 200 GB
    and the host >= 160 GB of memory (node store 32 GB + the oracle's copy + staging).  A probe that RAISES fails the test
    (round 4): a box that qualifies must never silently run the reduced size."""
    env = os.environ.get("FNV_FULLSIZE")
    if env is not None:
        return env == "1"
    import psutil
    import torch

    gpu, host = torch.cuda.get_device_properties(0).total_memory, psutil.virtual_memory().available
    SUMMARY_LINES.append("full-size probe: GPU %.0f GB (>= 200 needed), host memory available %.0f GB (>= 160 needed)"
                         % (gpu / 2 ** 30, host / 2 ** 30))
    return gpu >= 200 * 2 ** 30 and host >= 160 * 2 ** 30


@pytest.fixture(scope="module")
def hipmod():
    from flatnav_amd import hip

    assert hip.device_count() >= 1, "no MI355X visible"
    return hip


_CACHE = {}


def _oracle_index(oracle_mod, metric, X, M, efc):
    key = (metric, X.shape, M, efc, float(X[0, 0]), float(X[-1, -1]))
    if key not in _CACHE:
        ix = oracle_mod.OracleIndex.create(metric, X.shape[1], X.shape[0], M, "float32")
        ix.add(X, efc)  # one thread: the same graph in every run
        _CACHE[key] = ix
    return _CACHE[key]


def _float_parity(oracle_mod, hipmod, ix, Q, K, ef, kernels=("default", "two_heaps"), ref_bar=REF_ORDER_BAR):
    dev = hipmod.DeviceIndex.upload(ix.blob(), ix.node_size, ix.data_size, ix.M, ix.cur_nodes, ix.dtype, ix.metric, ix.dim)
    ix.use_reference_distance(False)
    od, ol, ost = ix.search(Q, K, ef, threads=8, stats=True)
    have_ref = ix.use_reference_distance(True)
    rd, rl = ix.search(Q, K, ef, threads=8) if have_ref else (None, None)
    ix.use_reference_distance(False)
    for kern in kernels:
        dev.set_option("sorted_beam", 0 if kern == "two_heaps" else 2)
        gd, gl, gst = dev.search(Q, K, ef, stats=True)
        same = (ol == gl).all(axis=1)
        print("%s kernel, ef=%d: ids identical to the oracle on %.2f%% of %d queries" % (kern, ef, 100 * same.mean(), len(Q)))
        SUMMARY_LINES.append("%s %dx%d ef=%d %s kernel: ids == oracle on %.2f%% of %d queries"
                             % (ix.metric, ix.cur_nodes, ix.dim, ef, kern, 100 * same.mean(), len(Q)))
        assert same.mean() >= ID_BAR, "%s: ids identical on only %.4f of the queries" % (kern, same.mean())
        assert np.allclose(od[same], gd[same], rtol=1e-5, atol=1e-6)
        # same path through the graph <=> same counters
        assert (ost["n_dist"][same] == gst["n_dist"][same]).mean() >= ID_BAR
        assert (np.diff(gd, axis=1) >= 0).all() and (gst["count"] == K).all()
        if have_ref:
            same_r = (rl == gl).all(axis=1)
            print("%s kernel vs the reference's distance kernel: ids identical on %.2f%%" % (kern, 100 * same_r.mean()))
            SUMMARY_LINES.append("%s %dx%d ef=%d %s kernel: ids == oracle on the reference's AVX-512 distances on %.2f%%"
                                 % (ix.metric, ix.cur_nodes, ix.dim, ef, kern, 100 * same_r.mean()))
            assert same_r.mean() >= ref_bar, "%s vs reference distance kernel: %.4f" % (kern, same_r.mean())
            assert np.allclose(rd[same_r], gd[same_r], rtol=1e-5, atol=1e-6)
    return dev


@pytest.mark.parametrize("kind", ["randn_unit", "lowrank_unit"])
def test_c3_shape_768d_float_inner_product_ef200(oracle_mod, hipmod, kind):
    N, NQ = (12000, 1000) if kind == "randn_unit" else (20000, 1000)
    if kind == "randn_unit":
        X, Q = ds.randn(N, NQ, 768, seed=768, normalize=True)
    else:
        X, Q = ds.lowrank_normalized(N, NQ, dim=768, rank=32, seed=7712)
    ix = _oracle_index(oracle_mod, "angular", X, 32, 100)
    dev = _float_parity(oracle_mod, hipmod, ix, Q, 10, 200, ref_bar=REF_ORDER_BAR_RANDN_768 if kind == "randn_unit" else REF_ORDER_BAR)
    dev.set_option("sorted_beam", 2)
    dev.search(Q, 10, 200)
    g = dev.launch_geometry()
    assert g["kernel"] == "merged_beam_registers"  # what C3 runs on by default
    if kind == "lowrank_unit":
        gt = ds.exact_topk_ip(X, Q, 10)
        _, gl = dev.search(Q, 10, 200)
        assert ds.recall_at_k(gl, gt) >= 0.95


@pytest.mark.parametrize("ef", [50, 100, 200, 400])
def test_c4_shape_100d_unit_vectors_ef_sweep(oracle_mod, hipmod, ef):
    X, Q = ds.lowrank_normalized(30000, 1000, dim=100, rank=24, seed=100)
    ix = _oracle_index(oracle_mod, "angular", X, 32, 100)
    _float_parity(oracle_mod, hipmod, ix, Q, 10, ef)


def test_c5_shape_128d_randn_l2(oracle_mod, hipmod):
    X, Q = ds.randn(40000, 1000, 128, seed=50)
    ix = _oracle_index(oracle_mod, "l2", X, 32, 100)
    _float_parity(oracle_mod, hipmod, ix, Q, 10, 100)


@pytest.mark.parametrize("config,n_small,n_full", [("c3-lowrank", 400_000, 10_000_000), ("c5", 2_000_000, 50_000_000),
                                                   ("c5-uint8", 2_000_000, 50_000_000)])
def test_fullsize_properties(oracle_mod, config, n_small, n_full):
    # Size-independent properties at the configurations' own sizes (see _fullsize), index built on the GPU from data
    # generated on the GPU: sortedness, exact recomputed distances for every returned id, idempotence across
    # launches, and GPU == oracle on a 1000-query sample of the same blob.
    import torch

    import flatnav_amd as flatnav

    full = _fullsize()
    N = n_full if full else n_small
    print("%s at N = %d (%s)" % (config, N, "FULL SIZE" if full else "reduced: not enough GPU / host memory, or FNV_FULLSIZE=0"))
    SUMMARY_LINES.append("test_fullsize_properties[%s]: N = %d x %d -- %s" % (config, N, 768 if config == "c3-lowrank" else 128,
                                                                            "FULL SIZE" if full else "REDUCED (box too small or FNV_FULLSIZE=0)"))
    dim, metric, ef = (768, "angular", 200) if config == "c3-lowrank" else (128, "l2", 100)
    dt = "uint8" if config == "c5-uint8" else "float32"  # c5-uint8 (round 6): an INTEGER dataset at HBM scale -- bit-exact ids
    NQ, K, M = 2000, 10, 32
    g = torch.Generator(device="cuda")
    g.manual_seed(7712 if config == "c3-lowrank" else 1296 if config == "c5-uint8" else 50)
    W = torch.randn((32, dim), generator=g, device="cuda") / 32 ** 0.5
    W16 = torch.randn((16, dim), generator=g, device="cuda") / 4

    def gen(m):
        if config == "c5":
            return torch.randn((m, dim), generator=g, device="cuda")
        if config == "c5-uint8":  # SURVEY 8d's S1 generator (bench.py Data._gen), stored as bytes
            x = 64 + 32 * (torch.randn((m, 16), generator=g, device="cuda") @ W16) + 6 * torch.randn((m, dim), generator=g, device="cuda")
            return torch.clip(torch.round(x), 0, 255).to(torch.uint8)
        x = torch.randn((m, 32), generator=g, device="cuda") @ W + 0.05 * torch.randn((m, dim), generator=g, device="cuda")
        return x / x.norm(dim=1, keepdim=True)

    ix = flatnav.index.create(metric, dim, N, M, getattr(flatnav.data_type.DataType, dt))
    ix.set_num_threads(min(16, os.cpu_count() or 1))
    chunk = 1_000_000 if dim > 256 else 5_000_000
    for first in range(0, N, chunk):
        ix.add(gen(min(chunk, N - first)).cpu().numpy(), 100, labels=list(range(first, min(N, first + chunk))), device=True)
    Q = gen(NQ).cpu().numpy()
    d, l = ix.search(Q, K, ef)
    assert (np.diff(d, axis=1) >= 0).all() and l.min() >= 0 and l.max() < N
    d2, l2 = ix.search(Q, K, ef)
    assert np.array_equal(d, d2) and np.array_equal(l, l2)
    blob = np.asarray(ix._raw_blob()).reshape(N, ix._node_size_bytes)
    if dt == "uint8":  # integer data: every returned distance is THE integer, ids / distances / counters equal the oracle's bits
        rows = blob[l.reshape(-1), :dim].astype(np.int64).reshape(NQ, K, dim)
        assert np.array_equal(((rows - Q[:, None, :].astype(np.int64)) ** 2).sum(-1).astype(np.float32), d)
        import ctypes

        from flatnav_amd import hip

        dev = hip.DeviceIndex(ctypes.c_void_p(ix.device_handle()), owned=False)
        gd, gl, gst = dev.search(Q, K, ef, stats=True)  # a full-grid launch (2000 queries, a shadow-free multi-slot one)
        o = oracle_mod.OracleIndex.from_blob(metric, dt, dim, N, N, M, blob.reshape(-1))
        od, ol, ost = o.search(Q[:1000], K, ef, threads=min(16, os.cpu_count() or 1), stats=True)
        assert np.array_equal(gl, l) and np.array_equal(ol, l[:1000]) and np.array_equal(od.view(np.uint32), d[:1000].view(np.uint32))
        assert np.array_equal(gst["n_dist"][:1000], ost["n_dist"]) and np.array_equal(gst["n_hops"][:1000], ost["n_hops"])
        SUMMARY_LINES.append("test_fullsize_properties[%s]: ids, distance bits, n_dist, n_hops == oracle on 1000 of 1000 queries at N = %d (uint8)" % (config, N))
        return
    rows = blob[l.reshape(-1), : dim * 4].copy().view(np.float32).reshape(NQ, K, dim)
    exact = ((rows - Q[:, None, :]) ** 2).sum(-1) if metric == "l2" else 1.0 - (rows * Q[:, None, :]).sum(-1)
    assert np.allclose(exact, d, rtol=1e-4, atol=1e-5)
    o = oracle_mod.OracleIndex.from_blob(metric, "float32", dim, N, N, M, blob.reshape(-1))
    od, ol = o.search(Q[:1000], K, ef, threads=min(16, os.cpu_count() or 1))
    same = (ol == l[:1000]).all(axis=1)
    print("%s N=%d: ids identical to the oracle on %.2f%% of 1000 queries" % (config, N, 100 * same.mean()))
    SUMMARY_LINES.append("test_fullsize_properties[%s]: GPU ids == oracle ids on %.2f%% of 1000 queries at N = %d" % (config, 100 * same.mean(), N))
    assert same.mean() >= ID_BAR and np.allclose(od[same], d[:1000][same], rtol=1e-5, atol=1e-6)
