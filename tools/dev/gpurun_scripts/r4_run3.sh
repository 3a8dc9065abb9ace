#!/bin/bash
# round 4, run 3: new tests (registers query, host pipeline, adopt, 8 shards); c3-lowrank steady state vs launch effects:
# 40 000-query launches (ramp / drain amortised) at several residencies, pure merged kernel vs 75 % exact tail, and the
# per-phase cycles of a loaded launch at 8 and 14 queries per CU; the c2 line with the pipelined host path
O=gpurun_out/r4_run3; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_round4.py tests/test_gpu_multi_device.py tests/test_gpu_round3.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
E=flatnav_amd/_exp
timeout 900 python tools/dev/knob_sweep.py --config c3-lowrank --ef 800 --rounds 2 --steps 3 --nq 40000 --nb 2 --no-tune \
  --libs r3=$E/libflatnav_hip_r3.so,w3p6=$E/lib768_w3p6.so,w2p8=$E/lib768_w2p8.so \
  --sets r3:sorted_variant=1 r3:sorted_variant=3 sorted_variant=1 sorted_variant=3 visited_slots=4096,sorted_variant=1 \
         visited_slots=1536,sorted_variant=1 visited_slots=1536,sorted_variant=3 visited_slots=768,sorted_variant=1 \
         w3p6:visited_slots=1536,sorted_variant=1 w3p6:visited_slots=4096,sorted_variant=1 w2p8:visited_slots=16384,sorted_variant=1 \
         w2p8:visited_slots=1536,sorted_variant=1 > $O/c3_40k.txt 2>$O/c3_40k.err
cat $O/c3_40k.txt
timeout 700 python tools/dev/knob_sweep.py --config c3-lowrank --ef 800 --rounds 1 --steps 3 --no-tune \
  --libs prof=$E/lib768_prof.so \
  --sets sorted_variant=1 sorted_variant=3 sorted_variant=0 prof:sorted_variant=1 prof:visited_slots=4096,sorted_variant=1 prof:visited_slots=1536,sorted_variant=1 \
         prof:visited_slots=16384,sorted_variant=1 prof:sorted_variant=0 > $O/c3_phases.txt 2>$O/c3_phases.err
cat $O/c3_phases.txt
timeout 300 python bench.py --secondary-configs none --steps 20 --warmup 5 > $O/bench_c2.json 2>$O/bench_c2.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4_run3/bench_c2.json"))
print(d["summary"], d["config"]["host_buffer_qps_pcie_inclusive"], d["config"]["recall_all_timed_batches"])
PY
