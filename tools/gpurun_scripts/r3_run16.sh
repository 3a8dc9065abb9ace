#!/bin/bash
# round 3, run 16: smoke() + the default bench line (with the pipelined object for every configuration) on the clean build
mkdir -p gpurun_out/r3_run16
python __graft_entry__.py --smoke > gpurun_out/r3_run16/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r3_run16/smoke.log; tail -4 gpurun_out/r3_run16/smoke.log
( time python bench.py --steps 20 --warmup 5 > gpurun_out/r3_run16/bench.json 2> gpurun_out/r3_run16/bench.err ) 2>> gpurun_out/r3_run16/bench.err
tail -3 gpurun_out/r3_run16/bench.err
python bench.py --dtype uint8 --steps 20 --warmup 5 > gpurun_out/r3_run16/bench_uint8.json 2>/dev/null
