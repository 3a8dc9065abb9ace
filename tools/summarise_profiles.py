#!/usr/bin/env python3
"""Condenses gpurun_out/profile_set (tools/collect_profiles.sh) into the tracked files under profiles/."""
import csv, collections, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", "profile_set")
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
bench = json.load(open(os.path.join(O, "bench.json")))
shutil.copy(os.path.join(O, "bench.json"), os.path.join(P, "%s_bench_final.json" % tag))
shutil.copy(os.path.join(O, "trace", "bench_kernel_stats.csv"), os.path.join(P, "%s_rocprofv3_kernel_stats.csv" % tag))


def counters(d):
    rows = list(csv.DictReader(open(os.path.join(O, d, "bench_counter_collection.csv"))))
    rows = [r for r in rows if "beam_search" in r["Kernel_Name"]]
    big = max(int(r["Grid_Size"]) for r in rows)
    acc = collections.defaultdict(list)
    for r in rows:
        if int(r["Grid_Size"]) == big:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    meta = {k: rows[0][k] for k in ("Grid_Size", "Workgroup_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Kernel_Name")}
    meta["Grid_Size"] = str(big)
    return {k: sum(v) / len(v) for k, v in acc.items()}, meta


traffic = []
for ef in sorted({50, 60, bench["config"]["ef_search"]}):
    f, meta = counters("fetch_ef%d" % ef)
    w, _ = counters("write_ef%d" % ef)
    F, W = f["FETCH_SIZE"], w["WRITE_SIZE"]
    traffic.append({
        "n": 1000000, "nq": 10000, "ef": ef, "kernel": meta,
        "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) --output-format csv -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 --ef %d" % ef,
        "FETCH_SIZE_KB_per_launch": F, "WRITE_SIZE_KB_per_launch": W,
        "correction": "MI355X_MICROARCH.md HBM section: hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024; on gfx950 FETCH_SIZE tallies each 128-B request of a 16 B/lane coalesced read as 64 B -> read side doubled",
        "hbm_bytes_per_launch_uncorrected": (F + W) * 1024, "hbm_bytes_per_launch_corrected": (2 * F + W) * 1024,
    })
json.dump(traffic, open(os.path.join(P, "pmc_hbm_traffic.json"), "w"), indent=1)
sq, meta = counters("sq")
json.dump({"kernel": meta, "per_launch": sq, "ef": bench["config"]["ef_search"]}, open(os.path.join(P, "%s_sq_counters.json" % tag), "w"), indent=1)
print(json.dumps({"value": bench["value"], "ef": bench["config"]["ef_search"], "recall": bench["config"]["recall_at_10"],
                  "frac": bench["roofline"]["frac"], "avg_kernel_ms": bench["roofline"]["avg_kernel_ms"],
                  "cpu": bench["cpu_baseline"]["value"], "launch": bench["config"]["launch"]}))
print(open(os.path.join(P, "%s_rocprofv3_kernel_stats.csv" % tag)).read()[:420])
for t in traffic:
    print(t["ef"], "traffic GB", t["hbm_bytes_per_launch_corrected"] / 1e9)
print({k: round(v / 1e6, 1) for k, v in sq.items()})
