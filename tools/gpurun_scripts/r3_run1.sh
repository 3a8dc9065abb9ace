#!/bin/bash
# round 3, run 1: whole GPU suite on the new library (row stride, fnv_tune, views, threads per shard) + default bench
mkdir -p gpurun_out/r3_run1
python -m pytest tests -m gpu -x -q -s > gpurun_out/r3_run1/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_run1/pytest.log
tail -5 gpurun_out/r3_run1/pytest.log
( time python bench.py > gpurun_out/r3_run1/bench.json 2> gpurun_out/r3_run1/bench.err ) 2>> gpurun_out/r3_run1/bench.err
echo "bench rc=$?"
tail -c 1500 gpurun_out/r3_run1/bench.err
