"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE, see flatnav_oracle.cpp).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (flatnav_amd) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")
REF_PATH = os.path.join(HERE, "_ref", "libflatnav_ref.so")

METRIC = {"l2": 0, "angular": 1, "ip": 1}
DTYPE_ORD = {"float32": 9, "uint8": 0, "int8": 4}
ORD_DTYPE = {v: k for k, v in DTYPE_ORD.items()}


def build(force: bool = False) -> None:
    """Compile liboracle.so (and oracle/_ref when /root/reference is present)."""
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(
        os.path.join(HERE, "flatnav_oracle.cpp")
    ):
        subprocess.check_call(["make", "-C", HERE, "oracle"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/include/flatnav"):
        subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.orc_last_error.restype = C.c_char_p
        L.orc_blob.restype = C.c_void_p
        L.orc_blob.argtypes = [C.c_void_p]
        L.orc_distance.restype = C.c_float
        L.orc_distance.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64]
        L.orc_create.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_void_p)]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.orc_set_distance_fn.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_add.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_int, C.c_int,
                              C.POINTER(C.c_uint64)]
        L.orc_search.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int] + [
            C.c_void_p
        ] * 7
        L.orc_replay_neighbors.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                           C.POINTER(C.c_int32)]
        L.orc_replay_search.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_uint64, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_uint64),
                                        C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.orc_save.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_load.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
        L.orc_from_blob.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p,
                                    C.POINTER(C.c_void_p)]
        _lib = L
    return _lib


_ref = None


def ref_lib():
    """The reference's own distance code (oracle/_ref), or None when it was never built."""
    global _ref
    if _ref is None and os.path.exists(REF_PATH):
        R = C.CDLL(REF_PATH)
        for name in ("ref_l2_f32", "ref_l2_u8", "ref_l2_i8", "ref_ip_f32", "ref_ip_u8", "ref_ip_i8",
                     "ref_default_l2_f32", "ref_default_ip_f32"):
            f = getattr(R, name)
            f.restype = C.c_float
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        R.ref_reduce_add8.restype = C.c_float
        R.ref_reduce_add8.argtypes = [C.c_void_p]
        R.ref_reduce_add4.restype = C.c_float
        R.ref_reduce_add4.argtypes = [C.c_void_p]
        R.ref_vs_new.restype = C.c_void_p
        R.ref_vs_new.argtypes = [C.c_uint32]
        for name in ("ref_vs_free", "ref_vs_clear"):
            getattr(R, name).argtypes = [C.c_void_p]
        R.ref_vs_insert.argtypes = [C.c_void_p, C.c_uint32]
        R.ref_vs_is_visited.argtypes = [C.c_void_p, C.c_uint32]
        R.ref_vs_mark.argtypes = [C.c_void_p]
        if hasattr(R, "ref_gorder"):  # round 6: Reordering.h, Multithreading.h, Datatype.h
            R.ref_gorder.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
            R.ref_gorder.restype = None
            R.ref_rcm.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
            R.ref_rcm.restype = None
            R.ref_execute_in_parallel.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
            R.ref_datatype_name.restype = C.c_char_p
            R.ref_datatype_name.argtypes = [C.c_int]
            R.ref_datatype_ordinal.argtypes = [C.c_char_p]
            for name in ("ref_datatype_size", "ref_datatype_ctype_bytes"):
                getattr(R, name).restype = C.c_uint64
                getattr(R, name).argtypes = [C.c_int]
            R.ref_datatype_enum_bytes.restype = C.c_uint64
        _ref = R
    return _ref


class OracleError(RuntimeError):
    pass


def _check(rc: int) -> None:
    if rc == 1:
        raise ValueError(lib().orc_last_error().decode())
    if rc != 0:
        raise OracleError(lib().orc_last_error().decode())


def _np_dtype(name: str):
    return {"float32": np.float32, "uint8": np.uint8, "int8": np.int8}[name]


class OracleIndex:
    """CPU oracle index with the reference's semantics (search, add, save/load)."""

    def __init__(self, handle, metric: str):
        self._h = handle
        self.metric = metric
        info = (C.c_uint64 * 8)()
        lib().orc_info(self._h, info)
        self.dtype = ORD_DTYPE[int(info[0])]
        self.M = int(info[1])
        self.data_size = int(info[2])
        self.node_size = int(info[3])
        self.max_nodes = int(info[4])
        self.dim = int(info[6])

    @classmethod
    def create(cls, metric: str, dim: int, max_nodes: int, M: int, dtype: str = "float32") -> "OracleIndex":
        h = C.c_void_p()
        _check(lib().orc_create(METRIC[metric], DTYPE_ORD[dtype], dim, max_nodes, M, C.byref(h)))
        return cls(h, metric)

    @classmethod
    def load(cls, path: str, metric: str) -> "OracleIndex":
        h = C.c_void_p()
        _check(lib().orc_load(path.encode(), METRIC[metric], C.byref(h)))
        return cls(h, metric)

    @classmethod
    def from_blob(cls, metric: str, dtype: str, dim: int, max_nodes: int, cur_nodes: int, M: int,
                  blob: np.ndarray) -> "OracleIndex":
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        h = C.c_void_p()
        _check(lib().orc_from_blob(METRIC[metric], DTYPE_ORD[dtype], dim, max_nodes, cur_nodes, M,
                                   blob.ctypes.data, C.byref(h)))
        return cls(h, metric)

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                lib().orc_free(self._h)
            except TypeError:  # interpreter shutdown: module globals are already gone
                pass
            self._h = None

    @property
    def cur_nodes(self) -> int:
        info = (C.c_uint64 * 8)()
        lib().orc_info(self._h, info)
        return int(info[5])

    def blob(self) -> np.ndarray:
        """Zero-copy view of the AoS node blob [max_nodes * node_size] (owned by the oracle)."""
        n = self.max_nodes * self.node_size
        ptr = lib().orc_blob(self._h)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n,))

    def use_reference_distance(self, enable: bool = True) -> bool:
        """Run the restated search on top of the reference's own compiled distance kernel."""
        R = ref_lib()
        if not enable or R is None:
            lib().orc_set_distance_fn(self._h, None)
            return False
        name = "ref_%s_%s" % ("l2" if METRIC[self.metric] == 0 else "ip",
                              {"float32": "f32", "uint8": "u8", "int8": "i8"}[self.dtype])
        fn = C.cast(getattr(R, name), C.c_void_p)
        lib().orc_set_distance_fn(self._h, fn)
        return True

    def add(self, data, ef_construction: int, num_initializations: int = 100, labels=None, threads: int = 1) -> int:
        data = np.ascontiguousarray(data, dtype=_np_dtype(self.dtype))
        if data.ndim != 2 or data.shape[1] != self.dim:
            raise ValueError("Data has incorrect dimensions.")
        lab_p = None
        if labels is not None:
            labels = np.ascontiguousarray(labels, dtype=np.int32)
            if labels.shape[0] != data.shape[0]:
                raise ValueError("Incorrect number of labels.")
            lab_p = labels.ctypes.data
        dc = C.c_uint64(0)
        _check(lib().orc_add(self._h, data.ctypes.data, data.shape[0], lab_p, ef_construction,
                             num_initializations, threads, C.byref(dc)))
        return int(dc.value)

    def search(self, queries, K: int, ef_search: int, num_initializations: int = 100, threads: int = 1,
               stats: bool = False):
        q = np.ascontiguousarray(queries, dtype=_np_dtype(self.dtype))
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise ValueError("Queries have incorrect dimensions.")
        nq = q.shape[0]
        d = np.empty((nq, K), dtype=np.float32)
        l = np.empty((nq, K), dtype=np.int32)
        cnt = np.empty(nq, dtype=np.int32)
        st = [np.zeros(nq, dtype=np.uint64) for _ in range(4)]
        _check(lib().orc_search(self._h, q.ctypes.data, nq, K, ef_search, num_initializations, threads,
                                d.ctypes.data, l.ctypes.data, cnt.ctypes.data,
                                *[s.ctypes.data for s in st]))
        if stats:
            return d, l, {"count": cnt, "n_dist": st[0], "n_hops": st[1], "n_admit": st[2], "max_cand": st[3]}
        return d, l

    def replay_search(self, query, K: int, ef_search: int, entry: int, log_d, log_ids, log_is_header):
        """The reference's search resumed from a traversal log (orc_replay_search: a design check for the GPU's mid-flight
        hand-over).  -> (dists[<=K], labels[<=K], n_dist, n_hops, hops taken from the log)"""
        q = np.ascontiguousarray(query, dtype=_np_dtype(self.dtype))
        d = np.ascontiguousarray(log_d, dtype=np.float32)
        i = np.ascontiguousarray(log_ids, dtype=np.uint32)
        hdr = np.ascontiguousarray(log_is_header, dtype=np.uint8)
        od, ol, cnt = np.empty(K, dtype=np.float32), np.empty(K, dtype=np.int32), C.c_int32(0)
        nd, nh, rep = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        _check(lib().orc_replay_search(self._h, q.ctypes.data, K, ef_search, entry, d.ctypes.data, i.ctypes.data, hdr.ctypes.data,
                                       len(d), od.ctypes.data, ol.ctypes.data, C.byref(cnt), C.byref(nd), C.byref(nh), C.byref(rep)))
        return od[:cnt.value], ol[:cnt.value], nd.value, nh.value, rep.value

    def save(self, path: str) -> None:
        _check(lib().orc_save(self._h, path.encode()))


def replay_neighbors(dists, ids, buffer_size: int, K: int):
    """The reference's neighbours heap and result assembly (Index.h:693-704, 393-408) driven by a log of evaluated
    neighbours -- entry point first, then (distance, node id) in evaluation order.  Returns (float32[<=K], uint32[<=K])."""
    d = np.ascontiguousarray(dists, dtype=np.float32)
    i = np.ascontiguousarray(ids, dtype=np.uint32)
    od, oi, cnt = np.empty(K, dtype=np.float32), np.empty(K, dtype=np.uint32), C.c_int32(0)
    _check(lib().orc_replay_neighbors(d.ctypes.data, i.ctypes.data, len(d), buffer_size, K, od.ctypes.data, oi.ctypes.data,
                                      C.byref(cnt)))
    return od[:cnt.value], oi[:cnt.value]


def distance(metric: str, x: np.ndarray, y: np.ndarray) -> float:
    dt = str(x.dtype)
    x = np.ascontiguousarray(x)
    y = np.ascontiguousarray(y, dtype=x.dtype)
    return float(lib().orc_distance(METRIC[metric], DTYPE_ORD[dt], x.ctypes.data, y.ctypes.data, x.shape[0]))
