#!/bin/bash
mkdir -p gpurun_out/r3_run7
python tools/overlap_ab.py f32 > gpurun_out/r3_run7/overlap_f32.txt 2>&1
python tools/overlap_ab.py u8 > gpurun_out/r3_run7/overlap_u8.txt 2>&1
cat gpurun_out/r3_run7/*.txt
