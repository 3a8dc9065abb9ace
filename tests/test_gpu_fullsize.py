"""Full-size check at BASELINE.json's configs[1] shape (1M x 128 float32 L2, M=32, efc=100, ef=100, K=10,
10 000 queries): size-independent properties over the whole batch + GPU == CPU oracle on a sample the oracle
finishes in seconds + the recall bar of the metric."""
import os

import numpy as np
import pytest

from flatnav_amd import datasets as ds

pytestmark = pytest.mark.gpu


def test_sift1m_shape_properties_and_parity(oracle_mod):
    import torch

    import flatnav_amd as flatnav

    N, NQ, K, EF = 1_000_000, 10_000, 10, 100
    X, Q = ds.sift_like(N, NQ)
    index = flatnav.index.create("l2", 128, N, 32)
    index.set_num_threads(min(24, os.cpu_count() or 1))
    index.add(X, 100)
    d, l = index.search(Q, K, EF)
    # (1) shape / order / range
    assert d.shape == (NQ, K) and l.shape == (NQ, K) and (np.diff(d, axis=1) >= 0).all()
    assert l.min() >= 0 and l.max() < N and all(len(set(r)) == K for r in l[:2000].tolist())
    # (2) every returned distance is the exact distance to the returned id (integer-valued data: bit-exact)
    diff = X[l.reshape(-1)].reshape(NQ, K, 128) - Q[:, None, :]
    assert np.array_equal((diff * diff).sum(-1, dtype=np.float32), d)
    # (3) idempotence: a second launch over the same slots returns the same bytes
    d2, l2 = index.search(Q, K, EF)
    assert np.array_equal(d, d2) and np.array_equal(l, l2)
    # (4) GPU == CPU oracle (ids, distances) on a 500-query sample, including counters through the C ABI
    o = oracle_mod.OracleIndex.from_blob("l2", "float32", 128, N, N, 32, np.asarray(index._raw_blob()))
    od, ol, ost = o.search(Q[:500], K, EF, threads=16, stats=True)
    assert np.array_equal(ol, l[:500]) and np.array_equal(od, d[:500])
    import ctypes

    from flatnav_amd import hip

    dev = hip.DeviceIndex(ctypes.c_void_p(index.device_handle()), owned=False)
    gd, gl, gst = dev.search(Q[:500], K, EF, stats=True)
    assert np.array_equal(gl, ol) and np.array_equal(gd.view(np.uint32), od.view(np.uint32))
    assert np.array_equal(gst["n_dist"], ost["n_dist"]) and np.array_equal(gst["n_hops"], ost["n_hops"])
    # ... and in the FULL launch (10 000 queries over the resident slots: rounds, hand-overs, the tuned variant)
    fd, fl, fst = dev.search(Q, K, EF, stats=True)
    assert np.array_equal(fl, l) and np.array_equal(fd.view(np.uint32), d.view(np.uint32))
    assert np.array_equal(fst["n_dist"][:500], ost["n_dist"]) and np.array_equal(fst["n_hops"][:500], ost["n_hops"])
    # (5) the metric's recall bar, against exact brute force on the GPU
    xt, qt = torch.from_numpy(X).cuda(), torch.from_numpy(Q[:1000]).cuda()
    xn = (xt * xt).sum(1)
    gt = torch.cat([torch.topk(xn[None, :] - 2.0 * (qt[s:s + 250] @ xt.T), K, dim=1, largest=False).indices
                    for s in range(0, 1000, 250)]).cpu().numpy()
    assert ds.recall_at_k(l[:1000], gt) >= 0.95


def test_rccl_replication_path_single_rank(tmp_path):
    # The multi-GPU load path (device-buffer views + torch.distributed broadcast, backend nccl == RCCL) with a
    # 1-rank group: validates the zero-copy views of the C-ABI buffers and that a replica searches identically.
    import subprocess
    import sys

    code = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29581", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from flatnav_amd import hip, multigpu, datasets as ds
from oracle import oracle as orc
X, Q = ds.sift_like(3000, 200)
ix = orc.OracleIndex.create("l2", 128, 3000, 16); ix.add(X, 64)
dev = hip.DeviceIndex.upload(ix.blob(), ix.node_size, ix.data_size, 16, 3000, "float32", "l2", 128)
multigpu.replicate_index(dev, 0, src=0)
replica = hip.DeviceIndex.alloc(16, 3000, "float32", "l2", 128)
for (src, n), (dst, _) in zip(dev.device_buffers(), replica.device_buffers()):   # stand-in for the peer's copy
    a = torch.as_tensor(multigpu._DevView(src, n), device="cuda:0"); b = torch.as_tensor(multigpu._DevView(dst, n), device="cuda:0")
    b.copy_(a)
torch.cuda.synchronize()
r1, r2 = dev.search(Q, 10, 64), replica.search(Q, 10, 64)
assert np.array_equal(r1[0], r2[0]) and np.array_equal(r1[1], r2[1])
od, ol = ix.search(Q, 10, 64)
assert np.array_equal(ol, r2[1])
dist.destroy_process_group()
print("REPLICA_OK")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "REPLICA_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
