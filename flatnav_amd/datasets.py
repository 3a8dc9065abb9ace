"""Synthetic dataset generators used by bench.py and the tests.

No dataset can be downloaded in this environment, so the SIFT-1M / GloVe / randn
configurations of BASELINE.json are reproduced by seeded generators with the
distributions that SURVEY.md section 8(d) specifies.  The streams are defined by THIS
code (chunked generation consumes the random stream in a different order than one
big call, see sift_like), so BASELINE.md's CPU numbers -- taken on the survey's own
one-call generators -- are statistically, not bytewise, comparable.
"""
from __future__ import annotations

import numpy as np


def sift_like(n: int, nq: int, dim: int = 128, rank: int = 16, seed: int = 1296):
    """S1 "int-lowrank" SIFT stand-in: integer-valued float32 in 0..255.

    Integer-valued data makes every L2 distance an exact integer < 2^24, so all
    summation orders agree bit-for-bit (CPU SIMD, oracle, GPU wavefront tree).
    Base and queries come from the same stream, base first.
    """
    rng = np.random.default_rng(seed)
    W = rng.standard_normal((rank, dim), dtype=np.float32) / 4

    def gen(m, chunk=200_000):
        out = np.empty((m, dim), dtype=np.float32)
        for s in range(0, m, chunk):
            e = min(m, s + chunk)
            z = rng.standard_normal((e - s, rank), dtype=np.float32)
            x = 64 + 32 * (z @ W) + 6 * rng.standard_normal((e - s, dim), dtype=np.float32)
            out[s:e] = np.clip(np.rint(x), 0, 255)
        return out

    # NOTE: chunked generation consumes the stream in a different order than a
    # single call would for n > chunk; the generator is defined by THIS code.
    X = gen(n)
    Q = gen(nq)
    return X, Q


def lowrank_normalized(n: int, nq: int, dim: int = 768, rank: int = 32, seed: int = 7712, noise: float = 0.05):
    """S3: low-intrinsic-dimension unit vectors for the angular / inner-product configs."""
    rng = np.random.default_rng(seed)
    W = rng.standard_normal((rank, dim), dtype=np.float32) / np.sqrt(np.float32(rank))

    def gen(m, chunk=100_000):
        out = np.empty((m, dim), dtype=np.float32)
        for s in range(0, m, chunk):
            e = min(m, s + chunk)
            z = rng.standard_normal((e - s, rank), dtype=np.float32)
            x = z @ W + noise * rng.standard_normal((e - s, dim), dtype=np.float32)
            x /= np.linalg.norm(x, axis=1, keepdims=True)
            out[s:e] = x
        return out

    return gen(n), gen(nq)


def randn(n: int, nq: int, dim: int, seed: int, normalize: bool = False):
    """Isotropic Gaussian data (configs C3/C5 as worded; adversarial for graph ANN)."""
    rng = np.random.default_rng(seed)

    def gen(m, chunk=200_000):
        out = np.empty((m, dim), dtype=np.float32)
        for s in range(0, m, chunk):
            e = min(m, s + chunk)
            x = rng.standard_normal((e - s, dim), dtype=np.float32)
            if normalize:
                x /= np.linalg.norm(x, axis=1, keepdims=True)
            out[s:e] = x
        return out

    return gen(n), gen(nq)


def exact_topk_l2(X: np.ndarray, Q: np.ndarray, k: int, block: int = 1024) -> np.ndarray:
    """Brute-force ground truth ids (squared L2), float64 accumulate-free formulation on float32 blocks."""
    out = np.empty((Q.shape[0], k), dtype=np.int64)
    xn = (X.astype(np.float64) ** 2).sum(1)
    for s in range(0, Q.shape[0], block):
        q = Q[s:s + block].astype(np.float64)
        d = xn[None, :] - 2.0 * (q @ X.T.astype(np.float64))
        idx = np.argpartition(d, k, axis=1)[:, :k]
        dd = np.take_along_axis(d, idx, 1)
        out[s:s + block] = np.take_along_axis(idx, np.argsort(dd, axis=1, kind="stable"), 1)
    return out


def exact_topk_ip(X: np.ndarray, Q: np.ndarray, k: int, block: int = 1024) -> np.ndarray:
    out = np.empty((Q.shape[0], k), dtype=np.int64)
    for s in range(0, Q.shape[0], block):
        d = 1.0 - Q[s:s + block].astype(np.float64) @ X.T.astype(np.float64)
        idx = np.argpartition(d, k, axis=1)[:, :k]
        dd = np.take_along_axis(d, idx, 1)
        out[s:s + block] = np.take_along_axis(idx, np.argsort(dd, axis=1, kind="stable"), 1)
    return out


def recall_at_k(found: np.ndarray, truth: np.ndarray) -> float:
    """Mean |returned intersect truth| / k (experiments/plotting/metrics.py:53-66 of the reference)."""
    k = truth.shape[1]
    hits = 0
    for f, t in zip(found, truth):
        hits += len(set(f[:k].tolist()) & set(t.tolist()))
    return hits / (k * len(truth))


def effective_cpus() -> int:
    """CPUs this process can really use: the scheduler affinity capped by the cgroup CPU quota
    (cpu.max = "<quota> <period>").  On the MI355X boxes 256 CPUs are visible but the quota is 16."""
    import math
    import os

    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, math.ceil(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            pr = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, math.ceil(q / pr)))
        except (OSError, ValueError):
            pass
    return n
