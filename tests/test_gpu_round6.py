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
