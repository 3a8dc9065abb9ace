#!/bin/bash
# round 5, run 15: one LDS round trip per admission into a full beam (coop_admit_full): parity, A/B against the round-4 library,
# timeline of the hand-overs, small batches
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r5_run15
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round5.py tests/test_golden.py tests/test_gpu_round3.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -n 'FAILED\|passed\|failed' $O/pytest.log | tail -8
for C in "c2 float32" "c2 uint8"; do set -- $C
timeout 900 python tools/dev/knob_sweep.py --config $1 --dtype $2 --ef 52 --rounds 3 --steps 10 --nb 8 --libs r4=flatnav_amd/_exp/libflatnav_hip_r4.so --sets \
  "base" "r4:base" "sorted_variant=0" "r4:sorted_variant=0" "sorted_variant=1" "sorted_variant=5" "sorted_variant=3" "sorted_variant=4" \
  > $O/sweep_$1_$2.txt 2>&1; echo "sweep $1 $2 rc=$?"; grep "^ef=" $O/sweep_$1_$2.txt | cut -c1-150
done
timeout 600 python tools/dev/launch_timeline.py --config c2 --dtype float32 --ef 52 --lib tl=flatnav_amd/_exp/libflatnav_hip_tl.so --variants=1,-1 > $O/timeline_c2_float32.txt 2>&1
grep -A6 "^## variant" $O/timeline_c2_float32.txt | cut -c1-200
timeout 900 python tools/dev/latency_probe.py 1000000 > $O/latency.txt 2>&1; grep -v "caller threads\|amdgpu" $O/latency.txt | head -24
