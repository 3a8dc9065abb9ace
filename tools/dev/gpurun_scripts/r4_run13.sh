#!/bin/bash
# round 4, run 13: deep runs of the randomised parity sweeps on the final library (other seeds than run 10's)
O=gpurun_out/r4_run13; mkdir -p $O
FNV_FUZZ_TRIALS=4000 FNV_FUZZ_SEED=811 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "random_shapes" > $O/fuzz_search.log 2>&1; tail -3 $O/fuzz_search.log
FNV_FUZZ_TRIALS=400 FNV_FUZZ_SEED=812 timeout 1500 python -m pytest tests/test_gpu_device_build.py tests/test_gpu_python_api.py -m gpu -x -q -k "random" > $O/fuzz_build.log 2>&1; tail -3 $O/fuzz_build.log
