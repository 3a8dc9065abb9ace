#!/bin/bash
# round 5, run 3: hand-over parity (fixed assertions), then launch timelines of the merged-beam kernel with and without the replay
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r5_run3
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_round5.py -q -m gpu -s > $O/pytest_a.log 2>&1; echo "pytest A rc=$?"; grep -n 'resumed from\|FAILED\|passed\|failed' $O/pytest_a.log | tail -20
for C in "c2 float32" "c2 uint8"; do set -- $C
for RP in 0 1; do
timeout 600 python tools/dev/launch_timeline.py --config $1 --dtype $2 --ef 52 --lib tl=flatnav_amd/_exp/libflatnav_hip_tl.so --variants=1,-1 --opt tie_replay=$RP > $O/timeline_$1_$2_replay$RP.txt 2>&1; echo "timeline $1 $2 replay=$RP rc=$?"
grep -A7 "^## variant" $O/timeline_$1_$2_replay$RP.txt
done; done
