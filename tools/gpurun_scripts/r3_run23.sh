#!/bin/bash
# round 3, run 23: deep randomised sweeps on the FINAL kernels (wide-tag probe rewrite, early link row), then the final default bench lines
mkdir -p gpurun_out/r3_run23
O=gpurun_out/r3_run23
FNV_FULLSIZE=0 FNV_FUZZ_TRIALS=500 FNV_FUZZ_SEED=77 python -m pytest tests/test_gpu_parity.py tests/test_gpu_device_build.py tests/test_gpu_python_api.py -m gpu -q -k "random" > $O/fuzz.log 2>&1
echo "fuzz rc=$?" >> $O/fuzz.log; tail -3 $O/fuzz.log
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; wc -c $O/bench.json
python bench.py --dtype uint8 --steps 20 --warmup 5 > $O/bench_uint8.json 2>/dev/null
