#!/usr/bin/env python3
"""Developer tool: re-runs the randomised merged-beam sweep of tests/test_gpu_parity.py and dumps the first mismatch."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np
import flatnav_amd as flatnav
from flatnav_amd import hip
out = sys.argv[1]
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2026
trials = int(sys.argv[3]) if len(sys.argv) > 3 else 150
rng = np.random.default_rng(seed)
for trial in range(trials):
    dt = ["float32", "uint8", "int8"][trial % 3]
    metric = ["l2", "angular"][int(rng.integers(0, 2))]
    dim = int(rng.choice([8, 24, 32, 64, 100, 128, 200, 768]))
    M = int(rng.choice([4, 8, 16, 32, 48, 70]))
    N = int(rng.integers(800, 12000))
    spread = int(rng.choice([2, 4, 16, 120]))
    if dt == "int8":
        X = rng.integers(-spread, spread, (N, dim)).astype(np.int8); Q = rng.integers(-spread, spread, (256, dim)).astype(np.int8)
    elif dt == "uint8":
        X = rng.integers(0, 2 * spread, (N, dim)).astype(np.uint8); Q = rng.integers(0, 2 * spread, (256, dim)).astype(np.uint8)
    else:
        X = rng.integers(0, 2 * spread, (N, dim)).astype(np.float32); Q = rng.integers(0, 2 * spread, (256, dim)).astype(np.float32)
    kw = {} if dt == "float32" else {"index_data_type": getattr(flatnav.data_type.DataType, dt)}
    ix = flatnav.index.create(metric, dim, N, M, **kw)
    ix.set_num_threads(4)
    ix.add(X, 40, device=True)
    dev = hip.DeviceIndex(ctypes.c_void_p(ix.device_handle()), owned=False)
    shapes = [(1, int(rng.integers(1, 9))), (10, int(rng.integers(10, 65))), (int(rng.integers(1, 80)), int(rng.integers(65, 257))),
              (10, int(rng.integers(257, 700)))]
    if seed != 2026:  # deeper sweeps: K == ef (every beam member is a result), more small beams
        e1, e2 = int(rng.integers(2, 40)), int(rng.integers(40, 200))
        shapes += [(e1, e1), (e2, e2), (int(rng.integers(1, 6)), int(rng.integers(2, 12)))]
    for K, ef in shapes:
        dev.set_option("sorted_beam", 0)
        wd, wl, ws = dev.search(Q, K, ef, stats=True)
        dev.set_option("sorted_beam", 1)
        for regs in (1, 0):
            dev.set_option("beam_registers", regs)
            gd, gl, gs = dev.search(Q, K, ef, stats=True)
            bad = np.nonzero((ws["n_dist"] != gs["n_dist"]) | (ws["n_hops"] != gs["n_hops"]) | (wl != gl).any(axis=1) | (wd.view(np.uint32) != gd.view(np.uint32)).any(axis=1))[0]
            if len(bad):
                print("seed", seed, "trial", trial, dt, metric, "d", dim, "M", M, "N", N, "spread", spread, "K", K, "ef", ef, "regs", regs, dev.launch_geometry())
                for q in bad[:10]:
                    d1, l1, s1 = dev.search(Q[q:q + 1], K, ef, stats=True)
                    r = dev.replayed_queries()
                    print("  query", q, "exact n_dist/hops", ws["n_dist"][q], ws["n_hops"][q], "merged", gs["n_dist"][q], gs["n_hops"][q], "alone", s1["n_dist"][0], s1["n_hops"][0], r,
                          "ids", wl[q][:5], gl[q][:5], "d", wd[q][:3], gd[q][:3])
                ix.save(os.path.join(out, "fuzz_index.bin"))
                np.save(os.path.join(out, "fuzz_Q.npy"), Q); np.save(os.path.join(out, "fuzz_X.npy"), X)
                sys.exit(0)
print("seed", seed, "trials", trials, "no mismatch")
