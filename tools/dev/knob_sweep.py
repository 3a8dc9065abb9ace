#!/usr/bin/env python3
"""Developer tool: A/B of C-ABI option sets on ONE index in ONE process on ONE box (the discipline VERDICT r3 #9 asks
for: gains under 5 % are only believed from alternating runs in one process).

  python tools/dev/knob_sweep.py --config c3-lowrank [--n 10000000] --ef 700,800 \
         --sets "base" "visited_slots=1536" "visited_slots=1536,sorted_cand_lds=0" [--rounds 2] [--recall]

Builds the configuration's index with bench.py's own generator and the device builder, then for every ef and every
option set: fnv_tune once, `--steps` timed launches over rotating query batches (HIP events on the launch stream), the
sets alternating `--rounds` times.  Prints one line per (ef, set): kernel ms (each round), queries/s of the best round,
algorithmic TB/s, launch geometry, chosen variant.  Results are never compared with the oracle here (parity tests do that).
"""
import argparse
import ctypes
import json
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
ap.add_argument("--config", default="c3-lowrank", choices=sorted(bench.CONFIGS))
ap.add_argument("--n", type=int, default=0)
ap.add_argument("--ef", default="")
ap.add_argument("--sets", nargs="*", default=["base"])
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--nb", type=int, default=4)
ap.add_argument("--nq", type=int, default=10_000)
ap.add_argument("--dtype", default="float32")
ap.add_argument("--recall", action="store_true")
ap.add_argument("--no-tune", action="store_true")
ap.add_argument("--json", default="")
ap.add_argument("--libs", default="", help="name=path,... further builds of libflatnav_hip.so to A/B against the in-tree one "
                                           "(each gets its own copy of the index: fnv_index_alloc + device-to-device copy)")
args = ap.parse_args()

cfg = dict(bench.CONFIGS[args.config])
N = args.n or cfg["n"]
NQ, NB, K, M, DIM = args.nq, args.nb, 10, 32, cfg["dim"]
DT = args.dtype
ESIZE = 4 if DT == "float32" else 1
dev_t = torch.device("cuda", 0)
torch.cuda.set_device(0)
t0 = time.time()
data = bench.Data(cfg, N, NQ * NB, torch, dev_t)
index = flatnav.index.create(distance_type=cfg["metric"], index_data_type=getattr(flatnav.data_type.DataType, DT), dim=DIM,
                             dataset_size=N, max_edges_per_node=M)
index.set_num_threads(16)
index.set_device(0)
rows = 1_000_000 if DIM > 256 else 5_000_000
for first, xh in data.chunks(rows):
    if DT == "uint8":
        xh = xh.astype(np.uint8)
    index.add(data=xh, ef_construction=100, labels=list(range(first, first + len(xh))), device=True)
print("# %s N=%d built in %.1fs" % (args.config, N, time.time() - t0), flush=True)
dev = hip.DeviceIndex(ctypes.c_void_p(index.device_handle()), owned=False)


def load_hip(path, name):
    """A second instance of flatnav_amd/hip.py bound to another build of the library (its own statics, its own handles)."""
    import importlib.util

    os.environ["FLATNAV_HIP_LIB"] = os.path.abspath(path)
    spec = importlib.util.spec_from_file_location("flatnav_amd.hip_" + name, os.path.join(ROOT, "flatnav_amd", "hip.py"))
    m = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = m
    spec.loader.exec_module(m)
    del os.environ["FLATNAV_HIP_LIB"]
    return m


from flatnav_amd import multigpu  # noqa: E402

dev._lib = hip.lib
DEVS = {"main": dev}
for item in [x for x in args.libs.split(",") if x]:
    name, path = item.split("=")
    m = load_hip(path, name)
    if hasattr(m.lib(), "fnv_index_adopt"):  # the same buffers, no copy
        d2 = m.DeviceIndex.adopt(dev.device_buffers(), M, N, DT, cfg["metric"], DIM, device=0, keep_alive=index)
    else:  # an older build: its own copy
        d2 = m.DeviceIndex.alloc(M, N, DT, cfg["metric"], DIM, device=0)
        for (sp, sb), (dp, db) in zip(dev.device_buffers(), d2.device_buffers()):
            nbytes = min(sb, db)
            torch.as_tensor(multigpu._DevView(dp, nbytes), device=dev_t).copy_(torch.as_tensor(multigpu._DevView(sp, nbytes), device=dev_t))
        torch.cuda.synchronize()
    d2._lib = m.lib
    DEVS[name] = d2
    print("# lib %s = %s (%s)" % (name, path, m.lib().fnv_version().decode()), flush=True)
Q = data.queries()
if DT == "uint8":
    Q = Q.astype(np.uint8)
dq = torch.from_numpy(np.ascontiguousarray(Q).reshape(NB, NQ, DIM)).to(dev_t)
od = torch.empty((NQ, K), dtype=torch.float32, device=dev_t)
ol = torch.empty((NQ, K), dtype=torch.int32, device=dev_t)
nd = torch.zeros(NQ, dtype=torch.int64, device=dev_t)
nh = torch.zeros(NQ, dtype=torch.int64, device=dev_t)
stream = torch.cuda.current_stream()
gt = bench.exact_topk(torch, dev, dq[0], K, N, DIM, DT, cfg["metric"]) if args.recall else None
step_nodes = max(1, N // 100)
n_scan = (N + step_nodes - 1) // step_nodes


def split(spec):
    """'lib:opt=v,opt=v' -> (handle, option string); no 'lib:' = the in-tree library."""
    name, _, opts = spec.rpartition(":")
    return DEVS[name or "main"], opts


def apply(spec, undo=False):
    d, opts = split(spec)
    if opts == "base":
        return
    for kv in opts.split(","):
        k, v = kv.split("=")
        try:
            d.set_option(k, DEFAULTS[k] if undo else int(v))
        except ValueError:  # an option an older build of the library does not know: ignored when undoing
            if not undo:
                raise


DEFAULTS = dict(visited_factor=27, visited_slots=0, visited_floor=2048, occupancy_target=13, occupancy_roomy=9, cand_factor=2,
                cand_slots=0, blocks_per_cu=0, sorted_beam=2, sorted_cand_lds=2, sorted_tail_exact_pct=-1, beam_registers=1,
                sorted_variant=-1, tune_layout=1, shadow_exact=1, entry_kernel=0, visited_tag_bits=0, overflow_list=-1,
                wide_table_max=-1, query_regs=1, tie_replay=1, tie_log_entries=0)


def timed(dev, ef, steps):
    evs = []
    for i in range(steps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        dev.search_device(dq[i % NB].data_ptr(), NQ, K, ef, 100, od.data_ptr(), ol.data_ptr(), 0, nd.data_ptr(), nh.data_ptr(),
                          stream=stream.cuda_stream)
        b.record(stream)
        evs.append((a, b))
    torch.cuda.synchronize()
    dev.status()
    return float(np.mean([a.elapsed_time(b) for a, b in evs]))


out = []
efs = [int(x) for x in args.ef.split(",")] if args.ef else [cfg["ef"] or 100]
for ef in efs:
    res = {}
    for rnd in range(args.rounds):
        for spec in args.sets:
            try:
                dev = split(spec)[0]
                apply(spec)
                if not args.no_tune:
                    dev.tune(int(dq[0].data_ptr()), K, ef, 100, nq=NQ)
                timed(dev, ef, 2)
                ms = timed(dev, ef, args.steps)
                g = dev.launch_geometry()
                byts = float(((n_scan + nd.cpu().numpy()) * DIM * ESIZE + nh.cpu().numpy() * M * 4 + K * 4).sum())
                rec = None
                if gt is not None and rnd == 0:
                    dev.search_device(dq[0].data_ptr(), NQ, K, ef, 100, od.data_ptr(), ol.data_ptr(), stream=stream.cuda_stream)
                    torch.cuda.synchronize()
                    rec = float((ol.long().unsqueeze(2) == gt.unsqueeze(1)).any(dim=2).float().mean().item())
                r = res.setdefault(spec, dict(ms=[], bytes=byts, geom=g, variant=dev.launch_info()["variant"], recall=rec))
                if hasattr(dev, "handover_stats"):
                    r["handover"] = "%s %s" % (dev.replayed_queries(), dev.handover_stats())
                r["ms"].append(round(ms, 4))
                if hasattr(dev._lib(), "fnv_debug_phase_cycles"):  # a -DFNV_PHASE_TIMING build: shader cycles per phase
                    L = dev._lib()
                    L.fnv_debug_phase_cycles.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
                    buf = (ctypes.c_uint64 * 16)()
                    L.fnv_debug_phase_cycles(dev._h, buf)  # reset
                    timed(dev, ef, 2)
                    hops = float(nh.sum().item())
                    L.fnv_debug_phase_cycles(dev._h, buf)
                    cyc = np.array(list(buf), dtype=np.float64) / 2
                    names = ["setup", "entry", "select", "link_row", "visited", "distances", "merge", "finalize"]
                    r["phases"] = " ".join("%s %.0f" % (n, c / hops) for n, c in zip(names, cyc[:8])) + " | exact-search phases %.0f | total %.0f cycles/hop" % (cyc[8:].sum() / hops, cyc.sum() / hops)
            except Exception as e:  # a layout that does not fit, an unknown option of an older library ...
                res.setdefault(spec, dict(ms=[], error=str(e)))
            finally:
                apply(spec, undo=True)
    for spec in args.sets:
        r = res[spec]
        if not r["ms"]:
            print("ef=%d %-40s ERROR %s" % (ef, spec, r.get("error")), flush=True)
            continue
        best = min(r["ms"])
        print("ef=%d %-40s ms %s  best %.0f q/s  %.2f TB/s alg (%.3f of 8)  recall %s  per_cu %d lds %d vis %d cand %d %s var %s"
              % (ef, spec, r["ms"], NQ / best * 1e3, r["bytes"] / best / 1e9, r["bytes"] / best / 1e9 / 8.0,
                 "-" if r["recall"] is None else "%.4f" % r["recall"], r["geom"]["blocks_per_cu"], r["geom"]["lds_bytes"],
                 r["geom"]["visited_slots"], r["geom"]["cand_slots"], r["geom"]["kernel"], r["variant"]), flush=True)
        if r.get("handover"):
            print("      last launch: " + r["handover"], flush=True)
        if r.get("phases"):
            print("      cycles/hop: " + r["phases"], flush=True)
        out.append(dict(ef=ef, set=spec, **{k: v for k, v in r.items()}))
if args.json:
    json.dump(out, open(args.json, "w"), indent=1, default=str)
