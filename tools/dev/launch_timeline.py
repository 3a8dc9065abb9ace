#!/usr/bin/env python3
"""Developer tool: the SHAPE of one search launch -- how many query slots are busy, and how many queries finish, in every
slice of the launch's duration -- from per-query start / end clock readings (a -DFNV_TIMELINE build of the library: the
per-query counter outputs carry s_memrealtime readings, 100 MHz, instead of counts; see merged_beam.hpp).

  python tools/dev/launch_timeline.py --config c2 --ef 52 --lib tl=flatnav_amd/_exp/libflatnav_hip_tl.so [--variants=-1,1,4] [--nq 10000]

Builds the configuration's index with bench.py's generator and the device builder (the tree's library), lets the tree's
library tune the launch (layout + kernel variant), then runs the SAME launch through the timeline build on the same device
buffers (fnv_index_adopt) with the variant pinned to what the tree's library chose (or to each of --variants), and prints, per
launch: duration, ramp (first query start -> 90 % of the slots busy), the time the dispenser ran dry, drain (dispenser dry ->
end), slot-busy fraction over the whole launch, and a 20-slice table of busy slots / finished queries.
"""
import argparse
import ctypes
import os
import sys
import time

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
ap.add_argument("--ef", type=int, default=52)
ap.add_argument("--nq", type=int, default=10_000)
ap.add_argument("--dtype", default="float32")
ap.add_argument("--lib", required=True, help="name=path of the -DFNV_TIMELINE build")
ap.add_argument("--variants", default="-1", help="kernel variants to pin (-1: what the tree's library chose)")
ap.add_argument("--slices", type=int, default=20)
ap.add_argument("--opt", action="append", default=[], help="C-ABI option name=value for the timeline build's handle")
args = ap.parse_args()

cfg = dict(bench.CONFIGS[args.config])
N = args.n or cfg["n"]
NQ, K, M, DIM, DT = args.nq, 10, 32, cfg["dim"], args.dtype
dev_t = torch.device("cuda", 0)
torch.cuda.set_device(0)
t0 = time.time()
data = bench.Data(cfg, N, NQ * 2, torch, dev_t)
index = flatnav.index.create(distance_type=cfg["metric"], index_data_type=getattr(flatnav.data_type.DataType, DT), dim=DIM,
                             dataset_size=N, max_edges_per_node=M)
index.set_num_threads(16)
index.set_device(0)
for first, xh in data.chunks(1_000_000 if DIM > 256 else 5_000_000):
    if DT == "uint8":
        xh = xh.astype(np.uint8)
    index.add(data=xh, ef_construction=100, labels=list(range(first, first + len(xh))), device=True)
print("# %s N=%d %s built in %.1fs" % (args.config, N, DT, time.time() - t0), flush=True)
dev = hip.DeviceIndex(ctypes.c_void_p(index.device_handle()), owned=False)

import importlib.util  # noqa: E402

name, path = args.lib.split("=")
os.environ["FLATNAV_HIP_LIB"] = os.path.abspath(path)
spec = importlib.util.spec_from_file_location("flatnav_amd.hip_" + name, os.path.join(ROOT, "flatnav_amd", "hip.py"))
m = importlib.util.module_from_spec(spec)
sys.modules[spec.name] = m
spec.loader.exec_module(m)
del os.environ["FLATNAV_HIP_LIB"]
tl = m.DeviceIndex.adopt(dev.device_buffers(), M, N, DT, cfg["metric"], DIM, device=0, keep_alive=index)

Q = data.queries()
if DT == "uint8":
    Q = Q.astype(np.uint8)
dq = torch.from_numpy(np.ascontiguousarray(Q).reshape(2, NQ, DIM)).to(dev_t)
od = torch.empty((NQ, K), dtype=torch.float32, device=dev_t)
ol = torch.empty((NQ, K), dtype=torch.int32, device=dev_t)
nd = torch.zeros(NQ, dtype=torch.int64, device=dev_t)
nh = torch.zeros(NQ, dtype=torch.int64, device=dev_t)
stream = torch.cuda.current_stream()


def launch(d, b):
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    d.search_device(dq[b].data_ptr(), NQ, K, args.ef, 100, od.data_ptr(), ol.data_ptr(), 0, nd.data_ptr(), nh.data_ptr(), stream=stream.cuda_stream)
    e.record(stream)
    torch.cuda.synchronize()
    d.status()
    return a.elapsed_time(e)


dev.tune(int(dq[0].data_ptr()), K, args.ef, 100, nq=NQ)
ms_tree = min(launch(dev, i % 2) for i in range(6))
g = dev.launch_geometry()
chosen = dev.launch_info()["variant"]
names = ["two_heaps", "merged_beam", "merged_beam_tail50", "merged_beam_tail75", "merged_beam_tail100", "merged_beam_tail25", "merged_beam_tail_shadows"]
print("# tree's library: %.4f ms per launch, %d slots (%d per CU), table %d, variant %s" % (ms_tree, g["grid_blocks"], g["blocks_per_cu"], g["visited_slots"], chosen), flush=True)
# the timeline build runs the same layout: table size and heap home pinned to what the tuned launch used
tl.set_option("tune_layout", 0)
tl.set_option("visited_slots", int(g["visited_slots"]))
tl.set_option("sorted_cand_lds", 1 if g["cand_slots"] else 0)
for o in args.opt:
    k_, v_ = o.split("=")
    tl.set_option(k_, int(v_))
for v in [int(x) for x in args.variants.split(",")]:
    if v < 0:
        v = names.index(chosen) if chosen in names else 1
    tl.set_option("sorted_variant", v)
    launch(tl, 1)
    ms = launch(tl, 0)
    gt = tl.launch_geometry()
    start = nd.cpu().numpy().astype(np.uint64)
    endw = nh.cpu().numpy().astype(np.uint64)
    kind = (endw >> np.uint64(60)).astype(np.int64)
    end = (endw & np.uint64((1 << 60) - 1)).astype(np.int64)
    start = start.astype(np.int64)
    t_first = start.min()
    s_us, e_us = (start - t_first) / 100.0, (end - t_first) / 100.0  # 100 MHz
    dur = e_us.max()
    slots = int(gt["grid_blocks"])
    # busy slots over time: +1 at a start, -1 at an end
    ev = np.concatenate([np.stack([s_us, np.ones_like(s_us)], 1), np.stack([e_us, -np.ones_like(e_us)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    busy = np.cumsum(ev[:, 1])
    t_ramp = ev[np.argmax(busy >= 0.9 * slots), 0] if (busy >= 0.9 * slots).any() else float("nan")
    t_dry = s_us.max()  # the last query handed out
    busy_area = float(np.sum((e_us - s_us))) / (dur * slots)
    lat = e_us - s_us
    print("\n## variant %s (%d slots, %d per CU): launch %.4f ms by events, %.1f us from first start to last end" % (names[v], slots, gt["blocks_per_cu"], ms, dur))
    print("ramp: 90 %% of the slots busy after %.1f us; dispenser dry at %.1f us (%.0f %% of the launch); drain %.1f us (%.0f %%); slots busy %.1f %% of slot-time"
          % (t_ramp, t_dry, 100 * t_dry / dur, dur - t_dry, 100 * (dur - t_dry) / dur, 100 * busy_area))
    for k, label in ((0, "merged beam only"), (1, "searched twice (equal keys)"), (2, "straight to the exact search")):
        sel = kind == k
        if sel.any():
            print("  %-30s %5d queries, latency p50 %.1f us  p90 %.1f  max %.1f; started %.1f ... %.1f us" %
                  (label, int(sel.sum()), np.percentile(lat[sel], 50), np.percentile(lat[sel], 90), lat[sel].max(), s_us[sel].min(), s_us[sel].max()))
    last = np.argsort(e_us)[-5:]
    if hasattr(tl, "handover_stats"):
        print("  hand-overs: %s; by reason %s" % (tl.handover_stats(), tl.replayed_queries()))
    print("  last five to finish: " + ", ".join("%.1f us (%s, started %.1f)" % (e_us[i], ["merged", "twice", "exact"][kind[i]], s_us[i]) for i in last))
    print("| slice (us) | busy slots (mean) | of all | queries finished | rate (M queries/s) |\n|---|---|---|---|---|")
    edges = np.linspace(0, dur, args.slices + 1)
    for a, b in zip(edges[:-1], edges[1:]):
        # mean busy slots in [a, b): integrate the step function
        ov = np.clip(np.minimum(e_us, b) - np.maximum(s_us, a), 0, None).sum() / (b - a)
        fin = int(((e_us >= a) & (e_us < b)).sum()) + (1 if b == edges[-1] else 0) * int((e_us == b).sum())
        print("| %.0f - %.0f | %.0f | %.2f | %d | %.2f |" % (a, b, ov, ov / slots, fin, fin / (b - a)))
