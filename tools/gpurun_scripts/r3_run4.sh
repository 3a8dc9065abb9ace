#!/bin/bash
mkdir -p gpurun_out/r3_run4
python tools/tie_reasons.py > gpurun_out/r3_run4/tie_reasons.txt 2>&1
cat gpurun_out/r3_run4/tie_reasons.txt
