#!/bin/bash
# round 5, run 2: the hand-over log (replay instead of re-search): parity first, then A/B of pinned variants with the log on / off
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r5_run2
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_round5.py tests/test_gpu_bench.py -q -m gpu -s > $O/pytest_a.log 2>&1; echo "pytest A rc=$?"; grep -n 'resumed\|FAILED\|passed\|failed' $O/pytest_a.log | tail -20
for C in "c2 float32" "c2 uint8"; do set -- $C
timeout 600 python tools/dev/knob_sweep.py --config $1 --dtype $2 --ef 52 --rounds 3 --steps 10 --nb 8 --sets \
  "tie_replay=0" "tie_replay=1" "tie_replay=1,sorted_variant=1" "tie_replay=0,sorted_variant=1" "tie_replay=1,sorted_variant=5" "tie_replay=1,sorted_variant=2" "tie_replay=1,sorted_variant=4" \
  > $O/sweep_$1_$2.txt 2>&1; echo "sweep $1 $2 rc=$?"; grep -v "^\[" $O/sweep_$1_$2.txt | tail -12
done
