#!/bin/bash
# round 4, run 29: the final library (rounds 1-3's grid + the granule trim): launch-geometry-sensitive tests, then the bench lines
# of the two configurations the trim / grid question touched
O=gpurun_out/r4_run29; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_bench.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
timeout 200 python bench.py --config c2-uint8 --no-secondary --no-cpu-baseline > $O/bench_c2u8.json 2> $O/bench_c2u8.log; grep "^\[bench\]" $O/bench_c2u8.log | tail -1
timeout 400 python bench.py --config c3-lowrank --no-secondary --no-cpu-baseline > $O/bench_c3lr.json 2> $O/bench_c3lr.log; grep "^\[bench\]" $O/bench_c3lr.log | tail -1
