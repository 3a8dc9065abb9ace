#!/bin/bash
# round 4, run 13: final check of the committed tree -- the whole GPU suite, the default bench invocation, then deep runs of
# the randomised parity sweeps (other seeds than run 10's)
O=gpurun_out/r4_run13; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
( time python bench.py --gpus 1 ) > $O/bench.json 2> $O/bench.err
grep "^\[bench\]" $O/bench.err | tail -8
FNV_FUZZ_TRIALS=3000 FNV_FUZZ_SEED=811 timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "random_shapes" > $O/fuzz_search.log 2>&1; tail -2 $O/fuzz_search.log
FNV_FUZZ_TRIALS=300 FNV_FUZZ_SEED=812 timeout 1200 python -m pytest tests/test_gpu_device_build.py tests/test_gpu_python_api.py -m gpu -x -q -k "random" > $O/fuzz_build.log 2>&1; tail -2 $O/fuzz_build.log
