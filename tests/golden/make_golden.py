#!/usr/bin/env python3
"""Generates tests/golden/*.npz -- small index files + queries + expected search outputs.

PROVENANCE: these vectors are produced by the CPU ORACLE (oracle/flatnav_oracle.cpp, a restatement of
the reference's Index.h), NOT by an executed copy of the reference: flatnav/index/Index.h cannot be
compiled in this image (its `cereal` submodule is empty and stand-in headers are not allowed), and the
reference ships no golden vectors of its own.  They pin (a) the oracle against regressions, (b) the
on-disk format, (c) the GPU path on fixed bytes.  Re-run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from flatnav_amd import datasets as ds  # noqa: E402
from oracle import oracle as orc  # noqa: E402

CASES = {
    # name: (metric, dtype, N, dim, M, efc, generator)
    "l2_f32_siftlike": ("l2", "float32", 1500, 128, 16, 64, lambda rng, n, d: ds.sift_like(n, 1, dim=d, seed=77)[0]),
    "ip_f32_unit": ("angular", "float32", 1500, 100, 16, 64,
                    lambda rng, n, d: (lambda x: x / np.linalg.norm(x, axis=1, keepdims=True))(
                        rng.standard_normal((n, d)).astype(np.float32))),
    "l2_u8_ties": ("l2", "uint8", 2000, 16, 16, 64, lambda rng, n, d: rng.integers(0, 4, (n, d)).astype(np.uint8)),
    "ip_i8": ("angular", "int8", 1500, 37, 8, 48, lambda rng, n, d: rng.integers(-16, 16, (n, d)).astype(np.int8)),
}
SEARCHES = [(10, 50, 100), (1, 10, 7), (20, 100, 100)]  # (K, ef_search, num_initializations)


def main():
    orc.build()
    for name, (metric, dt, n, dim, M, efc, gen) in CASES.items():
        rng = np.random.default_rng(abs(hash(name)) % (2 ** 31) if False else sum(map(ord, name)))
        X = gen(rng, n, dim)
        Q = gen(np.random.default_rng(sum(map(ord, name)) + 1), 64, dim) if "siftlike" not in name else \
            ds.sift_like(n, 64, dim=dim, seed=77)[1]
        ix = orc.OracleIndex.create(metric, dim, n, M, dt)
        ix.add(X, efc)
        path = os.path.join(HERE, name + ".bin.tmp")
        ix.save(path)
        blob = np.frombuffer(open(path, "rb").read(), dtype=np.uint8)
        os.remove(path)
        out = {"index_file": blob, "queries": Q, "metric": np.array(metric), "dtype": np.array(dt),
               "dim": np.array(dim), "M": np.array(M), "n": np.array(n)}
        for K, ef, ninit in SEARCHES:
            d, l, st = ix.search(Q, K, ef, ninit, stats=True)
            key = "K%d_ef%d_init%d" % (K, ef, ninit)
            out[key + "_dist"] = d
            out[key + "_labels"] = l
            out[key + "_ndist"] = st["n_dist"]
            out[key + "_nhops"] = st["n_hops"]
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "index bytes", blob.size, "->", os.path.getsize(os.path.join(HERE, name + ".npz")))


if __name__ == "__main__":
    main()
