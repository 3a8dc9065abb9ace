#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run63; mkdir -p $O
show() {
  python - <<PY
import json
try:
    d=json.load(open("$1"))
    print(round(d["value"]), "ef", d["config"]["ef_search"], "recall", d["config"]["recall_at_10"], "frac", round(d["roofline"]["frac"],3), d["config"]["launch"]["kernel"], "reruns", d["config"]["queries_replayed_by_exact_kernel"], "sustained", round(d["sustained"]["value"]), "ms/step", round(d["ms_per_step"],3))
    for s in d["secondary"]: print("   ef", s["ef_search"], round(s["value"]), s["recall_at_10"], round(s["roofline_frac"],3))
    print("   cpu", round(d.get("cpu_baseline",{}).get("value",0)), d["config"]["index_build"], d["config"]["ef_selection"][-150:])
except Exception as e: print("FAILED", e)
PY
}
for cfg in "c2" "c2 --dtype uint8" "c4"; do
  tag=$(echo $cfg | tr ' =-' '___')
  timeout 900 python bench.py --config $cfg --steps 20 --warmup 5 > $O/bench_$tag.json 2> $O/bench_$tag.log; echo "bench $cfg rc=$?"
  show $O/bench_$tag.json
done
for cfg in "c3-lowrank" "c3" "c5"; do
  tag=$(echo $cfg | tr ' =-' '___')
  timeout 1500 python bench.py --config $cfg --steps 10 --warmup 3 > $O/bench_$tag.json 2> $O/bench_$tag.log; echo "bench $cfg rc=$?"
  tail -5 $O/bench_$tag.log
  show $O/bench_$tag.json
done
