"""N>1 logic on CPU: world_size-2 `gloo` processes exercise the replication + query-sharding protocol of
flatnav_amd.multigpu (broadcast of the three index buffers from rank 0, contiguous ceil(Q/G) shards, no
data-path collective), with the CPU oracle standing in for the per-rank device search."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_all_rows_once():
    from flatnav_amd.multigpu import shard_bounds

    for q in (0, 1, 7, 8, 9, 10000, 10001):
        for g in (1, 2, 3, 4, 8):
            rows = []
            for r in range(g):
                lo, hi = shard_bounds(q, g, r)
                assert 0 <= lo <= hi <= q and hi - lo <= (q + g - 1) // g
                rows += list(range(lo, hi))
            assert rows == list(range(q))
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    from flatnav_amd import datasets as ds
    from flatnav_amd import multigpu
    from oracle import oracle as orc

    dist.init_process_group("gloo", rank=rank, world_size=world)
    N, dim, M, Qn = 3000, 32, 16, 501
    X, Q = ds.sift_like(N, Qn, dim=dim)
    node = dim * 4 + 4 * M + 4
    # "device buffers": rank 0 owns the built index, the others an uninitialised replica of the same size
    if rank == 0:
        ix = orc.OracleIndex.create("l2", dim, N, M)
        ix.add(X, 64)
        blob = torch.from_numpy(ix.blob().copy())
    else:
        blob = torch.empty(N * node, dtype=torch.uint8)
    # split like the device does (three buffers) and replicate with one broadcast each
    bufs = [blob[: N * node // 3], blob[N * node // 3: 2 * (N * node // 3)], blob[2 * (N * node // 3):]]
    # ... in pieces of at most 10 000 bytes: the same code path that moves a 32 GB vector table as 16 collectives of 2 GB
    stats = multigpu.broadcast_buffers(bufs, src=0, piece_bytes=10_000)
    assert [st["bytes"] for st in stats] == [b.numel() for b in bufs]
    assert [st["pieces"] for st in stats] == [-(-b.numel() // 10_000) for b in bufs] and min(st["pieces"] for st in stats) > 1
    assert all(st["seconds"] > 0 and st["GBps"] > 0 for st in stats)
    replica = orc.OracleIndex.from_blob("l2", "float32", dim, N, N, M, blob.numpy())
    lo, hi = multigpu.shard_bounds(Qn, world, rank)
    d, l = replica.search(Q[lo:hi], 10, 50)
    full_l = multigpu.gather_rows(torch.from_numpy(l), Qn).numpy()
    full_d = multigpu.gather_rows(torch.from_numpy(d), Qn).numpy()
    if rank == 0:
        d1, l1 = ix.search(Q, 10, 50)  # single-"GPU" answer on the original index
        np.save(os.path.join(out_dir, "ok.npy"),
                np.array([np.array_equal(full_l, l1) and np.array_equal(full_d, d1), hi - lo]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_replicate_and_shard(tmp_path, oracle_mod):
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    ok = np.load(tmp_path / "ok.npy")
    assert ok[0] == 1 and ok[1] == 251  # ceil(501 / 2) rows on rank 0
