#!/bin/bash
# round 5, run 5: default bench invocation on the hand-over tree (+ the two bench tests that changed)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r5_run5
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_bench.py -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python bench.py --steps 20 --warmup 5 --full-record $O/bench_full.json > $O/bench_line.json 2> $O/bench.log; echo "bench rc=$?"
wc -c $O/bench_line.json
grep "^\[bench\] c" $O/bench.log | tail -7
python - <<PY
import json
d=json.load(open("$O/bench_full.json"))
for k in ["c2","c2-uint8","c4","c3-lowrank","c3","c5","c5-lowrank"]:
    e = d if k=="c2" else d[k]
    print(k, e["config"]["kernel_variant"], "tail", e["config"]["launch"]["tail_exact"], "replayed", e["config"]["queries_replayed_by_exact_kernel"], "pipelined", (e.get("pipelined") or {}).get("value"), "pcie", e["config"]["host_buffer_qps_pcie_inclusive"])
PY
