"""Dataset file readers for the formats flatnav's harness consumes (own implementation).

Same formats as the reference's experiments/data_loader.py:7-219 -- .npy; TEXMEX .fvecs/.ivecs/.bvecs
(each record: int32 dimension, then `dimension` elements); big-ann-benchmarks .fbin/.u8bin/.i8bin (header:
uint32 n, uint32 d, then n*d elements) and its ground-truth .bin (uint32 n, uint32 k, n*k uint32 ids,
n*k float32 distances).  Everything is memory-mapped; `rows=(start, stop)` slices without reading the rest.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import numpy as np

_VECS = {".fvecs": np.float32, ".ivecs": np.int32, ".bvecs": np.uint8}
_BIN = {".fbin": np.float32, ".u8bin": np.uint8, ".i8bin": np.int8}


def _slice(a: np.ndarray, rows: Optional[Tuple[int, int]]) -> np.ndarray:
    if rows is None:
        return a
    start, stop = rows
    if start < 0 or stop < start:
        raise ValueError("Invalid range specified: %r" % (rows,))
    return a[start:min(stop, a.shape[0])]


def read_vecs(path: str, rows: Optional[Tuple[int, int]] = None) -> np.ndarray:
    """TEXMEX *vecs file -> (n, d) array of the file's element type (a copy of the selected rows)."""
    ext = os.path.splitext(path)[1].lower()
    if ext not in _VECS:
        raise ValueError("not a .fvecs/.ivecs/.bvecs file: " + path)
    dt = np.dtype(_VECS[ext])
    dim = int(np.fromfile(path, dtype=np.int32, count=1)[0])
    rec = 4 + dim * dt.itemsize
    size = os.path.getsize(path)
    if dim <= 0 or size % rec:
        raise ValueError("corrupt vecs file: " + path)
    raw = np.memmap(path, dtype=np.uint8, mode="r").reshape(size // rec, rec)
    raw = _slice(raw, rows)
    return np.ascontiguousarray(raw[:, 4:]).view(dt).reshape(raw.shape[0], dim)


def read_bin(path: str, rows: Optional[Tuple[int, int]] = None) -> np.ndarray:
    """big-ann-benchmarks .fbin/.u8bin/.i8bin -> memory-mapped (n, d) array."""
    ext = os.path.splitext(path)[1].lower()
    if ext not in _BIN:
        raise ValueError("not a .fbin/.u8bin/.i8bin file: " + path)
    n, d = (int(x) for x in np.fromfile(path, dtype=np.uint32, count=2))
    a = np.memmap(path, dtype=_BIN[ext], mode="r", offset=8, shape=(n, d))
    return _slice(a, rows)


def read_ground_truth_bin(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """big-ann-benchmarks ground truth -> (ids uint32[n,k], distances float32[n,k])."""
    n, k = (int(x) for x in np.fromfile(path, dtype=np.uint32, count=2))
    ids = np.memmap(path, dtype=np.uint32, mode="r", offset=8, shape=(n, k))
    dist = np.memmap(path, dtype=np.float32, mode="r", offset=8 + 4 * n * k, shape=(n, k))
    return ids, dist


def load_matrix(path: str, rows: Optional[Tuple[int, int]] = None) -> np.ndarray:
    """Dispatch on the file extension (.npy / *vecs / *bin)."""
    ext = os.path.splitext(path)[1].lower()
    if not os.path.exists(path):
        raise FileNotFoundError("File %s not found" % path)
    if ext == ".npy":
        return _slice(np.load(path, mmap_mode="r"), rows)
    if ext in _VECS:
        return read_vecs(path, rows)
    if ext in _BIN:
        return read_bin(path, rows)
    raise ValueError("unsupported dataset format: " + path)


def load_dataset(train: str, queries: str, ground_truth: str, rows: Optional[Tuple[int, int]] = None,
                 normalize: bool = False):
    """(train, queries, ground-truth ids) as the reference's loaders return them; `normalize` applies the
    row normalisation its conversion script uses for angular datasets (convert_ann_benchmark_datasets.py:28-30)."""
    X = np.asarray(load_matrix(train, rows))
    Q = np.asarray(load_matrix(queries))
    gext = os.path.splitext(ground_truth)[1].lower()
    if gext == ".bin":
        G = np.asarray(read_ground_truth_bin(ground_truth)[0]).astype(np.int32)
    else:
        G = np.asarray(load_matrix(ground_truth)).astype(np.int32, copy=False)
    if normalize:
        X = X.astype(np.float32) / np.linalg.norm(X.astype(np.float32), axis=1, keepdims=True)
        Q = Q.astype(np.float32) / np.linalg.norm(Q.astype(np.float32), axis=1, keepdims=True)
    return X, Q, G
