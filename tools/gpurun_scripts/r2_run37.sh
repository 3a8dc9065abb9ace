#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run37; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "sorted_beam" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
show() {
  python - <<PY
import json
try:
    d=json.load(open("$1"))
    print(round(d["value"]), "ef", d["config"]["ef_search"], "recall", d["config"]["recall_at_10"], "frac", round(d["roofline"]["frac"],3), d["config"]["launch"]["kernel"], d["config"]["launch"]["tail_exact"], "reruns", d["config"]["queries_replayed_by_exact_kernel"], "sustained", round(d["sustained"]["value"]), "ms/step", round(d["ms_per_step"],3))
    for s in d["secondary"]: print("   ef", s["ef_search"], round(s["value"]), s["recall_at_10"], round(s["roofline_frac"],3))
except Exception as e: print("FAILED", e)
PY
}
for cfg in "c2" "c4" "c2 --dtype uint8"; do
  tag=$(echo $cfg | tr ' =-' '___')
  timeout 900 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$tag.json 2> $O/bench_$tag.log; echo "bench $cfg rc=$?"
  show $O/bench_$tag.json
done
