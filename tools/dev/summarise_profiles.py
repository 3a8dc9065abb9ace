#!/usr/bin/env python3
"""Condenses gpurun_out/profile_set (tools/dev/collect_profiles.sh) into the tracked files under profiles/:
  <tag>_bench.json                         the default bench invocation's line (every configuration)
  <tag>_kernel_trace_timed_regions.csv     per configuration: the timed region's search launches from rocprofv3's kernel trace
  <tag>_rocprofv3_kernel_stats_<cfg>.csv   rocprofv3 --stats of the same command (all launches of the process)
  <tag>_pmc_hbm_traffic.json               per configuration: FETCH_SIZE / WRITE_SIZE per launch, corrected as the guide prescribes
  <tag>_sq_counters.json                   instruction mix / wait counters, float32 headline and uint8 index"""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
O = os.path.join(ROOT, "gpurun_out", "profile_set")
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r6"
CONFIGS = ["c2", "c2-uint8", "c4", "c3-lowrank", "c3", "c5", "c5-lowrank", "c5-uint8"]


def find(d, suffix):
    hits = glob.glob(os.path.join(O, d, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


bench = json.load(open(os.path.join(O, "bench.json")))
if not (len(sys.argv) > 2 and sys.argv[1] == "--one"):
    shutil.copy(os.path.join(O, "bench.json"), os.path.join(P, "%s_bench.json" % tag))


def entry(c):
    return bench if c == "c2" else bench.get(c)


def timed_launches(c, steps):
    """The timed region of the profiled command = its last `steps` full-grid search launches (tuning, warm-up, the
    byte-count launches after the timed region and the recall checks launch the same kernels: bench.py's order is
    tune, warm-up, TIMED, counters, recall -- so the timed ones are found by position from the end)."""
    f = find("trace_" + c, "kernel_trace.csv")
    if not f:
        return None
    rows = [r for r in csv.DictReader(open(f)) if "beam_search" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    j = json.load(open(os.path.join(O, "trace_%s.json" % c)))
    # the launches of the timed region and everything bench.py launched after it have the geometry the line reports (the
    # device builder's and fnv_tune's other variants launch other grids); rocprofv3 counts work-items
    blocks = j["config"]["launch"]["grid_blocks"]
    full = [r for r in rows if int(r["Grid_Size_X"]) in (blocks, blocks * 64)]
    pos = j["roofline"]["trace_position"]  # counted by bench.py itself; the timed regions are back to back
    after, steps = pos["after"], pos["timed"] * pos.get("regions", 1)
    sel = full[-(after + steps):-after]
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel]
    r0 = sel[0]
    return {"config": c, "kernel": r0["Kernel_Name"], "calls": len(sel), "avg_ns": sum(dur) / len(dur), "min_ns": min(dur), "max_ns": max(dur),
            "grid": r0["Grid_Size_X"], "workgroup": r0["Workgroup_Size_X"], "lds": r0.get("LDS_Block_Size", ""), "vgpr": r0.get("VGPR_Count", ""),
            "sgpr": r0.get("SGPR_Count", ""), "scratch": r0.get("Scratch_Size", ""),
            "bench_events_avg_ms": j["roofline"]["avg_kernel_ms"], "algorithmic_bytes_per_launch": j["roofline"]["algorithmic_bytes_per_launch"],
            "ef": j["config"]["ef_search"]}


def counters(d, steps=3):
    f = find(d, "counter_collection.csv")
    if not f:
        return None, None
    after = None
    try:
        pos = json.load(open(os.path.join(O, d + ".json")))["roofline"]["trace_position"]
        after, steps = pos["after"], pos["timed"] * pos.get("regions", 1)
    except (OSError, KeyError, ValueError):
        pass
    rows = [r for r in csv.DictReader(open(f)) if "beam_search" in r["Kernel_Name"]]
    jd = os.path.join(O, d + ".json")
    if not os.path.exists(jd):
        jd = os.path.join(O, "fetch_" + d.split("_", 1)[1] + ".json")
    blocks = json.load(open(jd))["config"]["launch"]["grid_blocks"]
    rows = [r for r in rows if int(r["Grid_Size"]) in (blocks, blocks * 64)]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    if after is None:  # (a pass whose bench line was not kept: same command as the FETCH pass of the same configuration)
        pos = json.load(open(os.path.join(O, "fetch_" + d.split("_", 1)[1] + ".json")))["roofline"]["trace_position"]
        after, steps = pos["after"], pos["timed"] * pos.get("regions", 1)
    ids = ids[-(after + steps):-after]
    rows = [r for r in rows if int(r["Dispatch_Id"]) in ids]
    acc = collections.defaultdict(list)
    for r in rows:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    first = rows[0]
    meta = {k: first[k] for k in ("Grid_Size", "Workgroup_Size", "Scratch_Size", "VGPR_Count", "SGPR_Count", "Kernel_Name") if k in first}
    return {k: sum(v) / len(v) for k, v in acc.items()}, meta


ONE = None
if len(sys.argv) > 2 and sys.argv[1] == "--one":
    ONE, tag = sys.argv[2], "r5"
    CONFIGS = [ONE]
SUM = os.path.join(O, "summary")
os.makedirs(SUM, exist_ok=True)
if ONE is None and not any(find("trace_" + c, "kernel_trace.csv") for c in CONFIGS):
    # the raw rocprofv3 output was summarised on the GPU box, configuration by configuration (it is too big to come back)
    trace_rows, traffic, sq = [], [], {}
    for c in CONFIGS:
        f = os.path.join(SUM, c + ".json")
        if os.path.exists(f):
            part = json.load(open(f))
            trace_rows += part["trace"]
            traffic += part["traffic"]
            sq.update(part["sq"])
        st = os.path.join(SUM, "kernel_stats_%s.csv" % c)
        if os.path.exists(st):
            shutil.copy(st, os.path.join(P, "%s_rocprofv3_kernel_stats_%s.csv" % (tag, c)))
    CONFIGS = []
else:
    trace_rows, traffic, sq = [], [], {}
for c in CONFIGS:
    e = entry(c)
    if e is None:
        continue
    steps = 6 if c.startswith(("c3", "c5")) else 20
    t = timed_launches(c, steps)
    if t:
        trace_rows.append(t)
        st = find("trace_" + c, "kernel_stats.csv")
        if st:
            shutil.copy(st, os.path.join(SUM if ONE else P, ("kernel_stats_%s.csv" % c) if ONE else "%s_rocprofv3_kernel_stats_%s.csv" % (tag, c)))
    f, meta = counters("fetch_" + c)
    w, _ = counters("write_" + c)
    if f and w:
        F, W = f["FETCH_SIZE"], w["WRITE_SIZE"]
        fj = json.load(open(os.path.join(O, "fetch_%s.json" % c)))
        r = fj["roofline"]
        corrected = (2 * F + W) * 1024
        traffic.append({
            # (bench.recorded_traffic's key: the 1M uint8 index is recorded as config c2 / dtype uint8, every other one under its own name)
            "config": "c2" if c == "c2-uint8" else c, "dtype": "uint8" if c.endswith("uint8") else "float32",
            "n": int(fj["config"]["index_bytes_in_hbm"] // (r["row_stride_bytes"] + r.get("row_tail_bytes", 0) + 132)), "nq": 10000, "ef": fj["config"]["ef_search"], "kernel": meta,
            "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) --output-format csv -- python3 bench.py --config %s --ef %d "
                       "--no-cpu-baseline --no-secondary --sustain-seconds 0 --warmup 3 --regions 1 --steps 3" % (c, fj["config"]["ef_search"]),
            "FETCH_SIZE_KB_per_launch": F, "WRITE_SIZE_KB_per_launch": W,
            "correction": "MI355X_MICROARCH.md HBM section: hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024; on gfx950 FETCH_SIZE tallies each "
                          "128-B request of a 16 B/lane coalesced read as 64 B -> read side doubled",
            "hbm_bytes_per_launch_uncorrected": (F + W) * 1024, "hbm_bytes_per_launch_corrected": corrected,
            "algorithmic_bytes_per_launch": r["algorithmic_bytes_per_launch"], "line_bytes_per_launch": r["line_bytes_per_launch"],
            "traffic_over_algorithmic": corrected / r["algorithmic_bytes_per_launch"],
            "traffic_over_line_bytes": corrected / r["line_bytes_per_launch"],
            "kernel_ms_under_pmc": r["avg_kernel_ms"]})
    s, m = counters("sq_" + c)
    if s:
        sq[c] = {"kernel": m, "per_launch": s}
if ONE:
    json.dump({"trace": trace_rows, "traffic": traffic, "sq": sq}, open(os.path.join(SUM, ONE + ".json"), "w"), indent=1)
    print("summarised", ONE, [round(t["avg_ns"] / 1e6, 4) for t in trace_rows], [round(t["traffic_over_algorithmic"], 3) for t in traffic])
    sys.exit(0)
with open(os.path.join(P, "%s_kernel_trace_timed_regions.csv" % tag), "w") as fo:
    fo.write("# per configuration: the timed region's search launches of `rocprofv3 --kernel-trace --stats -- python3 bench.py --config <c> --ef <ef> "
             "--no-cpu-baseline --no-secondary --sustain-seconds 0 --warmup 3 --regions 1 --steps <calls>` (positions from the end of the process's full-grid launches)\n")
    cols = ["config", "ef", "kernel", "calls", "avg_ns", "min_ns", "max_ns", "grid", "workgroup", "lds", "vgpr", "sgpr", "scratch", "bench_events_avg_ms",
            "algorithmic_bytes_per_launch"]
    fo.write(",".join(cols) + ",achieved_GBps,frac_of_8TBps\n")
    for t in trace_rows:
        gbps = t["algorithmic_bytes_per_launch"] / t["avg_ns"]
        fo.write(",".join('"%s"' % t[k] if k == "kernel" else str(t[k]) for k in cols) + ",%.1f,%.4f\n" % (gbps, gbps / 8000.0))
json.dump(traffic, open(os.path.join(P, "%s_pmc_hbm_traffic.json" % tag), "w"), indent=1)
json.dump(sq, open(os.path.join(P, "%s_sq_counters.json" % tag), "w"), indent=1)
for t in trace_rows:
    e = entry(t["config"])
    print("%-11s ef=%-4d trace %.4f ms  bench-under-trace %.4f ms  default-run %.4f ms  frac(trace) %.3f  frac(default run) %.3f  scratch %s vgpr %s" % (
        t["config"], t["ef"], t["avg_ns"] / 1e6, t["bench_events_avg_ms"], e["roofline"]["avg_kernel_ms"],
        t["algorithmic_bytes_per_launch"] / t["avg_ns"] / 8000.0, e["roofline"]["frac"], t["scratch"], t["vgpr"]))
for t in traffic:
    print("%-11s %-7s ef=%-4d traffic %.2f GB = %.3f x algorithmic (%.3f x line bytes)  write %.2f GB" % (
        t["config"], t["dtype"], t["ef"], t["hbm_bytes_per_launch_corrected"] / 1e9, t["traffic_over_algorithmic"], t["traffic_over_line_bytes"],
        t["WRITE_SIZE_KB_per_launch"] * 1024 / 1e9))
