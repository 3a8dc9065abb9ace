#!/bin/bash
# round 5, run 9: the profile set of the hand-over tree (default bench; per configuration kernel trace + PMC passes; SQ counters)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
export PROFILE_TAG=r5
bash tools/dev/collect_profiles.sh > $R/gpurun_out/profile_set.log 2>&1
tail -30 $R/gpurun_out/profile_set.log
