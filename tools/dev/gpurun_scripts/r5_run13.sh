#!/bin/bash
# round 5, run 13: deep randomised parity sweeps on the hand-over tree (three seeds)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r5_run13
mkdir -p $O
cd $R
for SEED in 51 52 53; do
FNV_FUZZ_TRIALS=700 FNV_FUZZ_SEED=$SEED timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_device_build.py tests/test_gpu_python_api.py -q -m gpu -k "random" > $O/fuzz_$SEED.log 2>&1; echo "seed $SEED rc=$?"; tail -2 $O/fuzz_$SEED.log
done
