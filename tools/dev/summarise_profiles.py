#!/usr/bin/env python3
"""Condenses gpurun_out/profile_set (tools/collect_profiles.sh) into the tracked files under profiles/."""
import csv, collections, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
O = os.path.join(ROOT, "gpurun_out", "profile_set")
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r2"


def find(d, suffix):
    hits = glob.glob(os.path.join(O, d, "**", "*" + suffix), recursive=True)
    if not hits:
        raise SystemExit("no %s under %s" % (suffix, d))
    return hits[0]


bench = json.load(open(os.path.join(O, "bench.json")))
shutil.copy(os.path.join(O, "bench.json"), os.path.join(P, "%s_bench.json" % tag))
shutil.copy(os.path.join(O, "bench_uint8.json"), os.path.join(P, "%s_bench_uint8.json" % tag))
shutil.copy(find("trace", "kernel_stats.csv"), os.path.join(P, "%s_rocprofv3_kernel_stats.csv" % tag))
shutil.copy(find("trace_uint8", "kernel_stats.csv"), os.path.join(P, "%s_rocprofv3_kernel_stats_uint8.csv" % tag))


def bench_launches(d, out_name, steps=20):
    """The timed region's launches from the kernel trace: the last `steps` full-size (4096-slot) search launches.  The
    raw --stats table also averages the index construction's and the ef sweep's launches of the same kernels."""
    rows = [r for r in csv.DictReader(open(find(d, "kernel_trace.csv"))) if "beam_search" in r["Kernel_Name"]]
    full = [r for r in rows if int(r["Grid_Size_X"]) == max(int(x["Grid_Size_X"]) for x in rows)]
    last = sorted(full, key=lambda r: int(r["Start_Timestamp"]))[-steps:]
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last]
    names = collections.Counter(r["Kernel_Name"] for r in last)
    with open(os.path.join(P, out_name), "w") as f:
        f.write("# the last %d full-grid search launches of the profiled bench command (= its timed region), from rocprofv3's kernel trace\n" % steps)
        f.write("Kernel_Name,Calls,AverageNs,MinNs,MaxNs,Grid_Size,Workgroup_Size,LDS_Block_Size,VGPR_Count,SGPR_Count,Scratch_Size\n")
        for name, calls in names.items():
            sel = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last if r["Kernel_Name"] == name]
            r0 = next(r for r in last if r["Kernel_Name"] == name)
            f.write('"%s",%d,%.1f,%d,%d,%s,%s,%s,%s,%s,%s\n' % (name, calls, sum(sel) / len(sel), min(sel), max(sel), r0["Grid_Size_X"],
                    r0["Workgroup_Size_X"], r0.get("LDS_Block_Size", ""), r0.get("VGPR_Count", ""), r0.get("SGPR_Count", ""), r0.get("Scratch_Size", "")))
    return sum(dur) / len(dur) / 1e6


def counters(d, steps=3):
    """Per-launch averages over the profiled command's timed region: its last `steps` full-grid search launches (the
    index construction and the ef sweep launch search kernels too -- other instantiations, or the same one earlier)."""
    rows = [r for r in csv.DictReader(open(find(d, "counter_collection.csv"))) if "beam_search" in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})[-steps:]  # the timed steps are the command's last launches
    rows = [r for r in rows if int(r["Dispatch_Id"]) in ids]
    kernel = rows[-1]["Kernel_Name"]
    assert all(r["Kernel_Name"] == kernel for r in rows), "timed region mixes kernels"
    acc = collections.defaultdict(list)
    for r in rows:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    first = rows[0]
    meta = {k: first[k] for k in ("Grid_Size", "Workgroup_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Kernel_Name")}
    return {k: sum(v) / len(v) for k, v in acc.items()}, meta


avg_ms = bench_launches("trace", "%s_rocprofv3_bench_launches.csv" % tag)
avg_ms_u8 = bench_launches("trace_uint8", "%s_rocprofv3_bench_launches_uint8.csv" % tag)
print("timed-region launches under rocprofv3: float32 %.4f ms, uint8 %.4f ms (bench.py's own events: %.4f ms)" %
      (avg_ms, avg_ms_u8, bench["roofline"]["avg_kernel_ms"]))
traffic, sq = [], {}
for dt in ("float32", "uint8"):
    f, meta = counters("fetch_" + dt)
    w, _ = counters("write_" + dt)
    F, W = f["FETCH_SIZE"], w["WRITE_SIZE"]
    traffic.append({
        "config": "c2", "dtype": dt, "n": 1000000, "nq": 10000, "ef": bench["config"]["ef_search"], "kernel": meta,
        "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) --output-format csv -- python3 bench.py --no-cpu-baseline "
                   "--no-secondary --sustain-seconds 0 --ef %d --dtype %s --steps 3 --warmup 5" % (bench["config"]["ef_search"], dt),
        "FETCH_SIZE_KB_per_launch": F, "WRITE_SIZE_KB_per_launch": W,
        "correction": "MI355X_MICROARCH.md HBM section: hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024; on gfx950 FETCH_SIZE tallies each "
                      "128-B request of a 16 B/lane coalesced read as 64 B -> read side doubled",
        "hbm_bytes_per_launch_uncorrected": (F + W) * 1024, "hbm_bytes_per_launch_corrected": (2 * F + W) * 1024,
    })
    c, m = counters("sq_" + dt)
    sq[dt] = {"kernel": m, "per_launch": c}
for efw in (110, 200, 400):  # 100-d rows (config c4): 400-byte rows at a 512-byte stride since round 3
    try:
        wb = json.load(open(os.path.join(O, "bench_c4_ef%d.json" % efw)))
        f, meta = counters("fetch_c4_ef%d" % efw)
        w, _ = counters("write_c4_ef%d" % efw)
        F, W = f["FETCH_SIZE"], w["WRITE_SIZE"]
        r = wb["roofline"]
        corrected = (2 * F + W) * 1024
        traffic.append({"config": "c4", "dtype": "float32", "n": 1183514, "nq": 10000, "ef": efw, "kernel": meta,
                        "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py --config c4 --no-cpu-baseline "
                                   "--no-secondary --sustain-seconds 0 --ef %d --steps 3 --warmup 3" % efw,
                        "FETCH_SIZE_KB_per_launch": F, "WRITE_SIZE_KB_per_launch": W,
                        "hbm_bytes_per_launch_uncorrected": (F + W) * 1024, "hbm_bytes_per_launch_corrected": corrected,
                        "algorithmic_bytes_per_launch": r["algorithmic_bytes_per_launch"],
                        "line_bytes_per_launch": r["line_bytes_per_launch"],
                        "traffic_over_algorithmic": corrected / r["algorithmic_bytes_per_launch"],
                        "traffic_over_line_bytes": corrected / r["line_bytes_per_launch"],
                        "row_bytes": r["row_bytes"], "row_stride_bytes": r["row_stride_bytes"],
                        "note": "100-d float32 rows: 400 bytes of data in a 512-byte (four-line) stride; the write side is the visited "
                                "set's per-slot HBM bitmap (ids beyond the LDS table: read-modify-write of one line each)"})
    except (OSError, KeyError, SystemExit) as e:
        print("no c4 pass at ef=%d:" % efw, e)
json.dump([t for t in traffic if t["dtype"] == "float32"] + [t for t in traffic if t["dtype"] != "float32"],
          open(os.path.join(P, "%s_pmc_hbm_traffic.json" % tag), "w"), indent=1)
json.dump({"ef": bench["config"]["ef_search"], "counters": sq}, open(os.path.join(P, "%s_sq_counters.json" % tag), "w"), indent=1)
print(json.dumps({"value": bench["value"], "ef": bench["config"]["ef_search"], "recall": bench["config"]["recall_at_10"],
                  "frac": bench["roofline"]["frac"], "avg_kernel_ms": bench["roofline"]["avg_kernel_ms"],
                  "cpu": bench["cpu_baseline"]["value"], "launch": bench["config"]["launch"]}))
print(open(os.path.join(P, "%s_rocprofv3_kernel_stats.csv" % tag)).read()[:600])
for t in traffic:
    print(t["dtype"], "traffic GB", t["hbm_bytes_per_launch_corrected"] / 1e9, t["kernel"]["Kernel_Name"][:60])
for dt, v in sq.items():
    print(dt, {k: round(x / 1e6, 1) for k, x in v["per_launch"].items()})
