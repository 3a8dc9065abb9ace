"""ctypes binding of the C ABI in include/flatnav_hip.h (libflatnav_hip.so, gfx950).

This is the only way Python code reaches the device search path.  There is NO CPU fallback: if the
shared library is missing or no MI355X is visible, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FLATNAV_HIP_LIB") or os.path.join(HERE, "libflatnav_hip.so")

FNV_OK, FNV_ERR_INVALID, FNV_ERR_RUNTIME, FNV_ERR_NO_DEVICE, FNV_ERR_CAPACITY = 0, 1, 2, 3, 4
DTYPE_ORD = {"float32": 9, "uint8": 0, "int8": 4}
ORD_DTYPE = {v: k for k, v in DTYPE_ORD.items()}
METRIC_ORD = {"l2": 0, "angular": 1, "ip": 1}

# every symbol include/flatnav_hip.h declares (tests check the library exports exactly these)
C_ABI_SYMBOLS = [
    "fnv_last_error", "fnv_version", "fnv_device_count", "fnv_index_upload", "fnv_index_alloc",
    "fnv_index_device_buffers", "fnv_index_info", "fnv_index_free", "fnv_set_option", "fnv_search_batch",
    "fnv_search_batch_device", "fnv_search_status", "fnv_last_kernel_ms", "fnv_last_launch_geometry",
    "fnv_index_set_live_nodes", "fnv_index_write_nodes", "fnv_index_write_links", "fnv_index_insert_batch",
    "fnv_index_read_links", "fnv_last_replayed_queries", "fnv_replicate", "fnv_replica_refresh",
    "fnv_search_batch_multi", "fnv_index_view", "fnv_tune", "fnv_last_launch_info", "fnv_gather_ceiling",
    "fnv_index_adopt", "fnv_lane_info", "fnv_last_handover_stats", "fnv_row_layout",
]

_lib = None


class DeviceUnavailable(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load libflatnav_hip.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DeviceUnavailable(
            "flatnav_amd: %s is missing -- build it with `python -m flatnav_amd.build` "
            "(there is no CPU fallback for the search path)" % LIB_PATH)
    from ._runtime import preload_hip_runtime

    preload_hip_runtime()
    L = C.CDLL(LIB_PATH)
    L.fnv_last_error.restype = C.c_char_p
    L.fnv_version.restype = C.c_char_p
    L.fnv_device_count.argtypes = [C.POINTER(C.c_int)]
    L.fnv_index_upload.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_int, C.c_int,
                                   C.c_uint32, C.c_int, C.POINTER(C.c_void_p)]
    L.fnv_index_alloc.argtypes = [C.c_uint32, C.c_uint64, C.c_int, C.c_int, C.c_uint32, C.c_int,
                                  C.POINTER(C.c_void_p)]
    L.fnv_index_device_buffers.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.fnv_index_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    if hasattr(L, "fnv_row_layout"):  # (round 6; older builds under the A/B tools lack it)
        L.fnv_row_layout.argtypes = [C.c_uint32, C.c_int, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.fnv_index_free.argtypes = [C.c_void_p]
    L.fnv_index_set_live_nodes.argtypes = [C.c_void_p, C.c_uint64]
    L.fnv_index_write_nodes.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64]
    L.fnv_index_write_links.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    L.fnv_index_insert_batch.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
    L.fnv_index_read_links.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
    L.fnv_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.fnv_search_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5
    L.fnv_search_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int] + [
        C.c_void_p] * 6
    L.fnv_search_status.argtypes = [C.c_void_p]
    L.fnv_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L.fnv_last_replayed_queries.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.fnv_last_launch_geometry.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.fnv_index_view.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    if hasattr(L, "fnv_index_adopt"):  # (older builds of the library, A/B'ed by tools/dev/knob_sweep.py, lack it)
        L.fnv_index_adopt.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_int,
                                      C.c_uint32, C.c_int, C.POINTER(C.c_void_p)]
    L.fnv_replicate.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]
    L.fnv_replica_refresh.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    L.fnv_search_batch_multi.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_uint64, C.c_int, C.c_int,
                                         C.c_int] + [C.c_void_p] * 5
    L.fnv_tune.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int]
    L.fnv_last_launch_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.fnv_gather_ceiling.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
    # (round-5 entry points: an OLDER build of the library, loaded by the developer A/B tools through FLATNAV_HIP_LIB, lacks them)
    for name in ("fnv_lane_info", "fnv_last_handover_stats"):
        if hasattr(L, name) or LIB_PATH == os.path.join(HERE, "libflatnav_hip.so"):
            getattr(L, name).argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    _lib = L
    return L


def check(rc: int) -> None:
    """Map C-ABI status codes onto the exceptions the reference raises at the same points."""
    if rc == FNV_OK:
        return
    msg = lib().fnv_last_error().decode()
    if rc == FNV_ERR_INVALID:
        raise ValueError(msg)
    if rc == FNV_ERR_NO_DEVICE:
        raise DeviceUnavailable(msg)
    raise RuntimeError(msg)


def row_layout(dim: int, dtype: str, capacity: int):
    """(row_bytes, tail_bytes) of the vector table the library keeps for this geometry (fnv_row_layout; needs no GPU)."""
    rb, tb = C.c_uint32(0), C.c_uint32(0)
    check(lib().fnv_row_layout(dim, DTYPE_ORD[dtype], capacity, C.byref(rb), C.byref(tb)))
    return int(rb.value), int(tb.value)


def device_count() -> int:
    n = C.c_int(0)
    check(lib().fnv_device_count(C.byref(n)))
    return n.value


def _np_dtype(name: str):
    return {"float32": np.float32, "uint8": np.uint8, "int8": np.int8}[name]


def search_multi(indexes, queries, K: int, ef_search: int, num_initializations: int = 100, stats: bool = False):
    """One batch over several handles of the same index (replicas on several GPUs): rows [g*ceil(Q/G), ...) go to
    indexes[g], all devices work concurrently -> (dist float32[Q,K], labels int32[Q,K][, stats])."""
    first = indexes[0]
    q = np.ascontiguousarray(queries, dtype=_np_dtype(first.dtype))
    if q.ndim != 2 or q.shape[1] != first.dim:
        raise ValueError("Queries have incorrect dimensions.")
    nq = q.shape[0]
    d = np.empty((nq, K), dtype=np.float32)
    l = np.empty((nq, K), dtype=np.int32)
    cnt = np.empty(nq, dtype=np.int32)
    nd = np.zeros(nq, dtype=np.uint64)
    nh = np.zeros(nq, dtype=np.uint64)
    arr = (C.c_void_p * len(indexes))(*[ix._h for ix in indexes])
    check(lib().fnv_search_batch_multi(arr, len(indexes), q.ctypes.data, nq, K, ef_search, num_initializations,
                                       d.ctypes.data, l.ctypes.data, cnt.ctypes.data, nd.ctypes.data, nh.ctypes.data))
    if stats:
        return d, l, {"count": cnt, "n_dist": nd, "n_hops": nh}
    return d, l


class DeviceIndex:
    """An index resident in one GPU's HBM (vectors / links / labels in SoA form)."""

    def __init__(self, handle: C.c_void_p, owned: bool = True):
        """owned=False: a view of a handle that belongs to someone else (e.g. flatnav.index.*.device_handle());
        close() / garbage collection then leave it alone."""
        self._h = handle
        self._owned = owned
        info = (C.c_uint64 * 8)()
        check(lib().fnv_index_info(self._h, info))
        self.dtype = ORD_DTYPE[int(info[0])]
        self.M = int(info[1])
        self.row_bytes = int(info[2]) & 0xFFFFFFFF  # stride of the (main) vector table
        self.tail_bytes = int(info[2]) >> 32        # split rows (csrc/distance.hpp): bytes per row in the side table that follows it
        self.n_nodes = int(info[3])
        self.dim = int(info[4])
        self.metric = "l2" if int(info[5]) == 0 else "angular"
        self.device = int(info[6])

    @classmethod
    def upload(cls, blob: np.ndarray, node_size: int, data_size: int, M: int, n_nodes: int, dtype: str,
               metric: str, dim: int, device: int = 0) -> "DeviceIndex":
        blob = np.ascontiguousarray(blob).view(np.uint8).reshape(-1)
        if blob.size < node_size * n_nodes:
            raise ValueError("blob smaller than n_nodes * node_size")
        h = C.c_void_p()
        check(lib().fnv_index_upload(blob.ctypes.data, node_size, data_size, M, n_nodes, DTYPE_ORD[dtype],
                                     METRIC_ORD[metric], dim, device, C.byref(h)))
        return cls(h)

    @classmethod
    def alloc(cls, M: int, n_nodes: int, dtype: str, metric: str, dim: int, device: int = 0) -> "DeviceIndex":
        h = C.c_void_p()
        check(lib().fnv_index_alloc(M, n_nodes, DTYPE_ORD[dtype], METRIC_ORD[metric], dim, device, C.byref(h)))
        return cls(h)

    @classmethod
    def adopt(cls, buffers, M: int, n_nodes: int, dtype: str, metric: str, dim: int, device: int = 0, keep_alive=None) -> "DeviceIndex":
        """A handle on device buffers somebody else owns: `buffers` = [(ptr, nbytes)] * 3 as device_buffers() reports them
        (vectors at the library's row stride, links, labels).  `keep_alive`: whatever must outlive the handle."""
        h = C.c_void_p()
        check(lib().fnv_index_adopt(buffers[0][0], buffers[1][0], buffers[2][0], M, n_nodes, DTYPE_ORD[dtype],
                                    METRIC_ORD[metric], dim, device, C.byref(h)))
        v = cls(h)
        v._parent = keep_alive
        return v

    def close(self) -> None:
        if getattr(self, "_h", None):
            if getattr(self, "_owned", True):
                check(lib().fnv_index_free(self._h))  # fails (ValueError) while views of this handle are alive
            self._h = None
            self._parent = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_buffers(self):
        """[(ptr, nbytes)] * 3 for vectors, links, labels (for RCCL broadcast / peer copies)."""
        ptrs = (C.c_void_p * 3)()
        sizes = (C.c_uint64 * 3)()
        check(lib().fnv_index_device_buffers(self._h, ptrs, sizes))
        return [(int(ptrs[i] or 0), int(sizes[i])) for i in range(3)]

    # ---- incremental construction -------------------------------------------------------------
    def set_live_nodes(self, n_live: int) -> None:
        check(lib().fnv_index_set_live_nodes(self._h, int(n_live)))
        self.n_nodes = int(n_live)

    def write_nodes(self, first_node: int, rows: np.ndarray, node_size: int, data_size: int) -> None:
        """rows: uint8 blob of whole AoS node records for nodes first_node, first_node+1, ..."""
        rows = np.ascontiguousarray(rows).view(np.uint8).reshape(-1)
        if rows.size % node_size:
            raise ValueError("rows must hold whole node records")
        check(lib().fnv_index_write_nodes(self._h, int(first_node), rows.size // node_size, rows.ctypes.data,
                                          node_size, data_size))

    def write_links(self, node_ids: np.ndarray, link_rows: np.ndarray) -> None:
        ids = np.ascontiguousarray(node_ids, dtype=np.uint32).reshape(-1)
        rows = np.ascontiguousarray(link_rows, dtype=np.uint32).reshape(ids.size, self.M)
        check(lib().fnv_index_write_links(self._h, ids.ctypes.data, rows.ctypes.data, ids.size))

    def insert_batch(self, first_node: int, count: int, ef_construction: int, num_initializations: int = 100) -> int:
        """Search + wire nodes [first_node, first_node+count) on the device; returns the distance evaluations."""
        ev = C.c_uint64(0)
        check(lib().fnv_index_insert_batch(self._h, int(first_node), int(count), int(ef_construction),
                                           int(num_initializations), C.byref(ev)))
        self.n_nodes = int(first_node) + int(count)
        return int(ev.value)

    def read_links(self, first_node: int, count: int) -> np.ndarray:
        out = np.empty((int(count), self.M), dtype=np.uint32)
        check(lib().fnv_index_read_links(self._h, int(first_node), int(count), out.ctypes.data))
        return out

    def view(self) -> "DeviceIndex":
        """A second handle on the same device buffers with its own workspace: two searches in flight on one index."""
        h = C.c_void_p()
        check(lib().fnv_index_view(self._h, C.byref(h)))
        v = DeviceIndex(h)
        v._parent = self  # the view aliases this handle's buffers: keep it alive (the library refuses to free it first)
        return v

    def replicate(self, devices) -> list:
        """Replicas of this index on the given device ordinals (peer copies over xGMI); each is an independent handle."""
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        out = (C.c_void_p * len(devices))()
        check(lib().fnv_replicate(self._h, len(devices), devs, out))
        return [DeviceIndex(C.c_void_p(out[i])) for i in range(len(devices))]

    def refresh_replicas(self, replicas) -> None:
        arr = (C.c_void_p * len(replicas))(*[r._h for r in replicas])
        check(lib().fnv_replica_refresh(self._h, len(replicas), arr))
        for r in replicas:
            r.n_nodes = self.n_nodes

    def set_option(self, name: str, value: int) -> None:
        check(lib().fnv_set_option(self._h, name.encode(), int(value)))

    def search(self, queries, K: int, ef_search: int, num_initializations: int = 100, stats: bool = False):
        """Host-buffer batched search -> (dist float32[Q,K], labels int32[Q,K][, stats])."""
        q = np.ascontiguousarray(queries, dtype=_np_dtype(self.dtype))
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise ValueError("Queries have incorrect dimensions.")
        nq = q.shape[0]
        if K <= 0:
            raise ValueError("K must be positive")
        d = np.empty((nq, K), dtype=np.float32)
        l = np.empty((nq, K), dtype=np.int32)
        cnt = np.empty(nq, dtype=np.int32)
        nd = np.zeros(nq, dtype=np.uint64)
        nh = np.zeros(nq, dtype=np.uint64)
        check(lib().fnv_search_batch(self._h, q.ctypes.data, nq, K, ef_search, num_initializations, d.ctypes.data,
                                     l.ctypes.data, cnt.ctypes.data, nd.ctypes.data, nh.ctypes.data))
        if stats:
            return d, l, {"count": cnt, "n_dist": nd, "n_hops": nh}
        return d, l

    def search_into(self, queries: np.ndarray, K: int, ef_search: int, dist: np.ndarray, labels: np.ndarray,
                    num_initializations: int = 100) -> None:
        """Host-buffer batched search into arrays the CALLER owns (no allocation, no conversion): queries [Q, dim] of the index's
        element type, dist float32 [Q, K], labels int32 [Q, K], all C-contiguous.  When all three are pinned host memory
        (e.g. numpy views of torch tensors from pin_memory()) the call is zero-copy at any batch size."""
        if (queries.dtype != _np_dtype(self.dtype) or queries.ndim != 2 or queries.shape[1] != self.dim or not queries.flags.c_contiguous):
            raise ValueError("Queries have incorrect dimensions.")
        nq = queries.shape[0]
        if dist.shape != (nq, K) or labels.shape != (nq, K) or dist.dtype != np.float32 or labels.dtype != np.int32 \
                or not dist.flags.c_contiguous or not labels.flags.c_contiguous:
            raise ValueError("output arrays must be C-contiguous float32 / int32 [Q, K]")
        check(lib().fnv_search_batch(self._h, queries.ctypes.data, nq, K, ef_search, num_initializations, dist.ctypes.data,
                                     labels.ctypes.data, None, None, None))

    def search_device(self, q_ptr: int, nq: int, K: int, ef_search: int, num_initializations: int, dist_ptr: int,
                      label_ptr: int, count_ptr: int = 0, ndist_ptr: int = 0, nhops_ptr: int = 0,
                      stream: int = 0) -> None:
        """Device-buffer batched search, enqueued on `stream` (raw hipStream_t handle, 0 = null stream)."""
        check(lib().fnv_search_batch_device(self._h, q_ptr, nq, K, ef_search, num_initializations, dist_ptr,
                                            label_ptr, count_ptr or None, ndist_ptr or None, nhops_ptr or None,
                                            stream or None))

    def tune(self, queries, K: int, ef_search: int, num_initializations: int = 100, nq: int = 0) -> None:
        """Settle the adaptive kernel choice for (K, ef_search, this batch size) in one call (fnv_tune): afterwards no
        launch of that shape is an exploratory one.  `queries`: host array [Q, dim], or a device pointer (int) + nq."""
        if isinstance(queries, int):
            check(lib().fnv_tune(self._h, queries, int(nq), 1, K, ef_search, num_initializations))
            return
        q = np.ascontiguousarray(queries, dtype=_np_dtype(self.dtype))
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise ValueError("Queries have incorrect dimensions.")
        check(lib().fnv_tune(self._h, q.ctypes.data, q.shape[0], 0, K, ef_search, num_initializations))

    def launch_info(self) -> dict:
        """Variant of the most recent launch, whether it was an exploratory sample of the adaptive choice, and the host
        timestamps (ns, steady clock) of the most recent host-buffer search."""
        r = (C.c_uint64 * 4)()
        check(lib().fnv_last_launch_info(self._h, r))
        names = ["two_heaps", "merged_beam", "merged_beam_tail50", "merged_beam_tail75", "merged_beam_tail100",
                 "merged_beam_tail25", "merged_beam_tail_shadows"]
        return {"variant": names[int(r[0])], "variant_id": int(r[0]), "exploratory": bool(int(r[1]) & 1),
                "shadow": bool(int(r[1]) & 2),
                "enqueued_ns": int(r[2]), "completed_ns": int(r[3])}

    def gather_ceiling(self, waves_per_cu: int = 0) -> float:
        """GB/s of row bytes a pure random-row gather of this index's vector table reaches (fnv_gather_ceiling)."""
        v = C.c_double(0)
        check(lib().fnv_gather_ceiling(self._h, int(waves_per_cu), C.byref(v)))
        return float(v.value)

    def status(self) -> None:
        check(lib().fnv_search_status(self._h))

    def last_kernel_ms(self) -> float:
        ms = C.c_float(0)
        check(lib().fnv_last_kernel_ms(self._h, C.byref(ms)))
        return float(ms.value)

    def replayed_queries(self) -> dict:
        """Queries of the last search that the exact kernel replayed after the merged-beam kernel handed them over, by reason."""
        r = (C.c_uint64 * 5)()
        check(lib().fnv_last_replayed_queries(self._h, r))
        return dict(zip(["total", "eviction_tie", "selection_tie", "result_tie", "nan_inf"], [int(x) for x in r]))

    def handover_stats(self) -> dict:
        """Hand-overs of the last search (fnv_last_handover_stats): queries resumed from their log, hops taken from the logs,
        hops those queries' merged-beam passes had made, queries searched again from scratch."""
        r = (C.c_uint64 * 4)()
        if not hasattr(lib(), "fnv_last_handover_stats"):  # (an older build under the A/B tools)
            return dict(zip(["resumed", "hops_from_log", "hops_of_resumed", "from_scratch"], [0, 0, 0, self.replayed_queries()["total"]]))
        check(lib().fnv_last_handover_stats(self._h, r))
        return dict(zip(["resumed", "hops_from_log", "hops_of_resumed", "from_scratch"], [int(x) for x in r]))

    def launch_geometry(self) -> dict:
        g = (C.c_uint64 * 8)()
        check(lib().fnv_last_launch_geometry(self._h, g))
        keys = ["grid_blocks", "block_threads", "lds_bytes", "blocks_per_cu", "visited_slots", "cand_slots"]
        out = {k: int(g[i]) for i, k in enumerate(keys)}
        out["kernel"] = ["two_heaps", "merged_beam_registers", "merged_beam_lds"][int(g[6])]
        out["tail_exact"] = int(g[7])
        return out


def _lane_info(dev) -> list:
    r = (C.c_uint64 * 16)()
    check(lib().fnv_lane_info(dev._h, r))
    return [int(x) for x in r]


def lane_workspaces(dev) -> list:
    """Bytes of launch workspace (HBM) held by the handle [0] and by each of its hidden lanes [1..7] (fnv_lane_info)."""
    return _lane_info(dev)[0::2]


def lane_exploratory_launches(dev) -> list:
    """Exploratory launches of the adaptive kernel choice on the handle [0] and on each hidden lane [1..7] (always 0 there)."""
    return _lane_info(dev)[1::2]
