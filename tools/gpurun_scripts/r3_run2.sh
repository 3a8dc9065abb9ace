#!/bin/bash
# round 3, run 2: the whole GPU suite (full sizes by default); phase clocks of the uint8 kernel under load; uint8 layouts
mkdir -p gpurun_out/r3_run2
python -m pytest tests -m gpu -q -s > gpurun_out/r3_run2/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_run2/pytest.log
tail -15 gpurun_out/r3_run2/pytest.log
python tools/phase_profile.py --dtype uint8 --ef 52 > gpurun_out/r3_run2/phase_u8_ef52.txt 2>&1
python tools/phase_profile.py --dtype float32 --ef 52 > gpurun_out/r3_run2/phase_f32_ef52.txt 2>&1
for opts in "" "--opt sorted_cand_lds=0" "--opt sorted_cand_lds=0 --opt visited_slots=3072" "--opt blocks_per_cu=16" "--opt blocks_per_cu=12" "--opt sorted_variant=1"; do
  echo "== uint8 $opts" >> gpurun_out/r3_run2/u8_layouts.txt
  python bench.py --dtype uint8 --ef 52 --no-cpu-baseline --no-secondary --sustain-seconds 0 --steps 10 $opts 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['roofline']['avg_kernel_ms'], d['config']['launch'], d['config']['kernel_variant'], d['config']['queries_replayed_by_exact_kernel'])" >> gpurun_out/r3_run2/u8_layouts.txt 2>&1
done
cat gpurun_out/r3_run2/u8_layouts.txt
