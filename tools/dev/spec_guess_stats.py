#!/usr/bin/env python3
"""Developer probe: how often is the node a hop expands the RUNNER-UP that was guessed one hop ahead (merged_beam.hpp spec_node)?
That fraction bounds what any speculation on the runner-up (its link row today; its neighbours' distances, if one wanted to
software-pipeline the gather) can hide.  Needs a -DFNV_SPEC_STATS build of the library:

  python -c "from flatnav_amd import build; build.build(defines=['FNV_SPEC_STATS'], out='flatnav_amd/_exp/libflatnav_hip_spec.so')"
  python tools/dev/spec_guess_stats.py --lib flatnav_amd/_exp/libflatnav_hip_spec.so --config c2 --ef 52 [--dtype uint8] [--n N]
(the index is built by the in-tree library; the probe build adopts its buffers, like tools/dev/knob_sweep.py --libs)
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import flatnav_amd as flatnav  # noqa: E402
from flatnav_amd import hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c2", choices=sorted(bench.CONFIGS))
ap.add_argument("--n", type=int, default=0)
ap.add_argument("--ef", default="52")
ap.add_argument("--dtype", default="float32")
ap.add_argument("--nq", default="10000,64,1")
ap.add_argument("--lib", default=os.path.join(ROOT, "flatnav_amd", "_exp", "libflatnav_hip_spec.so"))
args = ap.parse_args()
cfg = dict(bench.CONFIGS[args.config])
N, DIM, DT = args.n or cfg["n"], cfg["dim"], args.dtype
dev_t = torch.device("cuda", 0)
data = bench.Data(cfg, N, 10000, torch, dev_t)
index = flatnav.index.create(distance_type=cfg["metric"], index_data_type=getattr(flatnav.data_type.DataType, DT), dim=DIM,
                             dataset_size=N, max_edges_per_node=32)
index.set_num_threads(16)
for first, xh in data.chunks(5_000_000):
    index.add(data=xh.astype(np.uint8) if DT == "uint8" else xh, ef_construction=100, labels=list(range(first, first + len(xh))), device=True)
own = hip.DeviceIndex(ctypes.c_void_p(index.device_handle()), owned=False)
import importlib.util

os.environ["FLATNAV_HIP_LIB"] = os.path.abspath(args.lib)
spec = importlib.util.spec_from_file_location("flatnav_amd.hip_spec", os.path.join(ROOT, "flatnav_amd", "hip.py"))
hip2 = importlib.util.module_from_spec(spec)
sys.modules[spec.name] = hip2
spec.loader.exec_module(hip2)
del os.environ["FLATNAV_HIP_LIB"]
L = hip2.lib()
if not hasattr(L, "fnv_debug_spec_stats"):
    raise SystemExit("this library build has no fnv_debug_spec_stats (build with -DFNV_SPEC_STATS, see the docstring)")
L.fnv_debug_spec_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
dev = hip2.DeviceIndex.adopt(own.device_buffers(), 32, N, DT, cfg["metric"], DIM, device=0, keep_alive=index)
Q = data.queries()
Q = Q.astype(np.uint8) if DT == "uint8" else Q
dev.set_option("sorted_beam", 1)
dev.set_option("sorted_variant", 1)  # every query in the merged-beam kernel
for ef in [int(x) for x in args.ef.split(",")]:
    for nq in [int(x) for x in args.nq.split(",")]:
        dev.search(Q[:nq], 10, ef)
        out = (ctypes.c_uint64 * 2)()
        hip2.check(L.fnv_debug_spec_stats(dev._h, out))
        print("%s %s N=%d ef=%d nq=%d: %d of %d hops expanded the guessed runner-up = %.1f %%" % (
            args.config, DT, N, ef, nq, out[0], out[1], 100.0 * out[0] / max(1, out[1])), flush=True)
