"""Multi-GPU layer: index replicated on every GPU of a node, queries sharded, no per-query collective.

One process per GPU (torch.distributed; backend "nccl" == RCCL over xGMI on ROCm).  The only
collective is the one-time replication of the three device index buffers from rank 0
(`replicate_index`), exactly what SURVEY.md 8(e) prescribes; searches are embarrassingly parallel over
query rows, like the reference's executeInParallel over rows (bindings.cpp:198-211).

The partitioning helpers are plain Python/torch and also run under the "gloo" backend on CPU tensors,
which is how tests/ exercise the N>1 logic without GPUs.
"""
from __future__ import annotations

from typing import List, Tuple


def shard_bounds(num_queries: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Rows [lo, hi) of the query matrix that `rank` answers: contiguous blocks of ceil(Q / G) rows
    (SURVEY.md 8e); trailing ranks may get fewer rows or none."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    per = (num_queries + world_size - 1) // world_size
    lo = min(num_queries, rank * per)
    return lo, min(num_queries, lo + per)


class _DevView:
    """Expose a raw device pointer to torch through the CUDA array interface (no copy)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


BROADCAST_PIECE_BYTES = 2 << 30  # one collective moves at most this much (a 32 GB vector table goes out as 16 pieces)


def broadcast_buffers(tensors: List["torch.Tensor"], src: int = 0, group=None, piece_bytes: int = BROADCAST_PIECE_BYTES,
                      sync=None) -> List[dict]:
    """Broadcast every buffer (vectors, links, labels) from `src`, in pieces of at most `piece_bytes`: one RCCL broadcast
    of 3.2e10 bytes is a single ring transfer nobody can watch, and its element count overflows a 32-bit int in more than
    one place of the stack; pieces cost nothing (each is still >= 1 GB: bandwidth-bound over xGMI) and give a progress and
    bandwidth report.  Works on CUDA tensors (RCCL) and on CPU tensors (gloo, used by the tests).  `sync`: a callable that
    waits for the device (bandwidth is measured per buffer when given).  Returns [{bytes, pieces, seconds, GBps}] per buffer."""
    import time

    import torch.distributed as dist

    stats = []
    for t in tensors:
        # (view, not reshape: on a non-contiguous tensor reshape would copy, and the non-source ranks would receive into a
        #  temporary while the caller's buffer stays unfilled -- view raises instead)
        flat = t.view(-1)
        per = max(1, piece_bytes // max(1, flat.element_size()))
        pieces = 0
        if sync:
            sync()
        t0 = time.perf_counter()
        for lo in range(0, flat.numel(), per):
            dist.broadcast(flat[lo:lo + per], src=src, group=group)
            pieces += 1
        if sync:
            sync()
        dt = time.perf_counter() - t0
        nbytes = flat.numel() * flat.element_size()
        stats.append({"bytes": nbytes, "pieces": pieces, "seconds": dt, "GBps": nbytes / dt / 1e9 if dt > 0 else None})
    return stats


def replicate_index(dev, device: int, src: int = 0, group=None, piece_bytes: int = BROADCAST_PIECE_BYTES) -> List[dict]:
    """Fill this rank's DeviceIndex (flatnav_amd.hip.DeviceIndex: uploaded on `src`, alloc()'ed elsewhere)
    from rank `src` with RCCL broadcasts of its three HBM buffers (in <= 2 GB pieces).  Returns broadcast_buffers' report,
    labelled vectors / links / labels."""
    import torch

    # broadcast the LIVE rows only: a device-built source may hold room for more nodes than it has, and the
    # replicas are allocated for the live count
    live = [dev.n_nodes * dev.row_bytes, dev.n_nodes * dev.M * 4, dev.n_nodes * 4]
    views, names = [], ["vectors", "links", "labels"]
    bufs = dev.device_buffers()
    for (ptr, nbytes), want in zip(bufs, live):
        if want > nbytes:
            raise RuntimeError("device buffer smaller than its live rows")
        views.append(torch.as_tensor(_DevView(ptr, want), device="cuda:%d" % device))
    tail = getattr(dev, "tail_bytes", 0)
    if tail:  # split rows: the side table follows the main table of THIS rank's capacity (csrc/beam_search.hip row_layout)
        ptr, nbytes = bufs[0]
        capacity = nbytes // (dev.row_bytes + tail)
        views.append(torch.as_tensor(_DevView(ptr + capacity * dev.row_bytes, dev.n_nodes * tail), device="cuda:%d" % device))
        names.append("vector tails")
    stats = broadcast_buffers(views, src=src, group=group, piece_bytes=piece_bytes, sync=lambda: torch.cuda.synchronize(device))
    for name, st in zip(names, stats):
        st["buffer"] = name
    return stats


def peer_access_matrix(n_devices: int) -> List[List[bool]]:
    """hipDeviceCanAccessPeer for every ordered pair of the first n devices (diagonal: True)."""
    import torch

    return [[i == j or bool(torch.cuda.can_device_access_peer(i, j)) for j in range(n_devices)] for i in range(n_devices)]


def gather_rows(local, num_queries: int, group=None):
    """Optional convenience: reassemble per-rank result blocks (ceil(Q/G)-row shards) on every rank.
    Not on the search path -- results normally stay with the rank that produced them."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    per = (num_queries + world - 1) // world
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat(parts, dim=0)[:num_queries]
