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


def broadcast_buffers(tensors: List["torch.Tensor"], src: int = 0, group=None) -> None:
    """One broadcast per buffer (vectors, links, labels).  Works on CUDA tensors (RCCL) and on CPU
    tensors (gloo, used by the tests)."""
    import torch.distributed as dist

    for t in tensors:
        dist.broadcast(t, src=src, group=group)


def replicate_index(dev, device: int, src: int = 0, group=None) -> None:
    """Fill this rank's DeviceIndex (flatnav_amd.hip.DeviceIndex: uploaded on `src`, alloc()'ed elsewhere)
    from rank `src` with RCCL broadcasts of its three HBM buffers."""
    import torch

    # broadcast the LIVE rows only: a device-built source may hold room for more nodes than it has, and the
    # replicas are allocated for the live count
    live = [dev.n_nodes * dev.row_bytes, dev.n_nodes * dev.M * 4, dev.n_nodes * 4]
    views = []
    for (ptr, nbytes), want in zip(dev.device_buffers(), live):
        if want > nbytes:
            raise RuntimeError("device buffer smaller than its live rows")
        views.append(torch.as_tensor(_DevView(ptr, want), device="cuda:%d" % device))
    broadcast_buffers(views, src=src, group=group)
    torch.cuda.synchronize(device)


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
