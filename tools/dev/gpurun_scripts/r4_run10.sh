#!/bin/bash
# round 4, run 10: the round's profile set from the final library (default bench invocation; per configuration a kernel
# trace of the timed region and FETCH_SIZE / WRITE_SIZE passes; SQ counters for the float32 headline and the uint8 index),
# then deep runs of the randomised parity sweeps on the same library
bash tools/dev/collect_profiles.sh > gpurun_out/profile_set.log 2>&1
tail -30 gpurun_out/profile_set.log
python tools/dev/summarise_profiles.py r4 > gpurun_out/profile_summary.txt 2>&1
cat gpurun_out/profile_summary.txt
O=gpurun_out/r4_run10; mkdir -p $O
FNV_FUZZ_TRIALS=1500 FNV_FUZZ_SEED=404 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "random_shapes" > $O/fuzz_search.log 2>&1; tail -3 $O/fuzz_search.log
FNV_FUZZ_TRIALS=200 FNV_FUZZ_SEED=405 timeout 900 python -m pytest tests/test_gpu_device_build.py tests/test_gpu_python_api.py -m gpu -x -q -k "random" > $O/fuzz_build.log 2>&1; tail -3 $O/fuzz_build.log
