#!/bin/bash
# round 5, run 7: the eager tail (merged-beam pass + early hand-over instead of straight to the exact search): parity, A/B per variant
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r5_run7
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round5.py tests/test_golden.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -n 'FAILED\|passed\|failed' $O/pytest.log | tail -8
for C in "c2 float32" "c2 uint8"; do set -- $C
timeout 900 python tools/dev/knob_sweep.py --config $1 --dtype $2 --ef 52 --rounds 3 --steps 10 --nb 8 --sets \
  "base" "tail_mode=0" "sorted_variant=5" "sorted_variant=2" "sorted_variant=3" "sorted_variant=4" "tail_mode=0,sorted_variant=3" "tail_mode=0,sorted_variant=4" "sorted_variant=1" \
  > $O/sweep_$1_$2.txt 2>&1; echo "sweep $1 $2 rc=$?"; grep -v "^\[\|amdgpu.ids" $O/sweep_$1_$2.txt | tail -20
done
