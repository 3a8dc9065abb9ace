#!/bin/bash
# round 3, run 8: device-built vs host-built graphs at 10M x 768 (S3) and 10M x 128 randn, 10 000 queries, exact ground truth
mkdir -p gpurun_out/r3_run8
python tools/builder_compare.py randn128 10000000 > gpurun_out/r3_run8/builder_randn128_10M.txt 2>&1
cat gpurun_out/r3_run8/builder_randn128_10M.txt
python tools/builder_compare.py lowrank768 10000000 > gpurun_out/r3_run8/builder_lowrank768_10M.txt 2>&1
cat gpurun_out/r3_run8/builder_lowrank768_10M.txt
