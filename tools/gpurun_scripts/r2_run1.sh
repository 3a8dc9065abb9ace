#!/bin/bash
# round-2 GPU run 1: parity suite + first bench lines of the reworked kernels
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run1; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -15 $O/pytest.log
for cfg in "c2" "c2 --opt sorted_beam=0" "c2 --dtype uint8" "c2 --dtype uint8 --opt register_beam=0" "c4"; do
  tag=$(echo $cfg | tr ' =-' '___')
  timeout 900 python bench.py --config $cfg --steps 20 --warmup 5 > $O/bench_$tag.json 2> $O/bench_$tag.log; echo "bench $cfg rc=$?"
  tail -c 1500 $O/bench_$tag.json | head -c 1500; echo
done
for bpc in 12 13 14 15; do
  timeout 600 python bench.py --config c2 --ef 50 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --opt blocks_per_cu=$bpc > $O/bench_bpc$bpc.json 2> $O/bench_bpc$bpc.log
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_bpc$bpc.json")); print("bpc $bpc", round(d["value"]), d["config"]["launch"])
except Exception as e: print("bpc $bpc failed", e)
PY
done
