#!/bin/bash
mkdir -p gpurun_out/r3_run6
python tools/layout_ab.py c4 64,110,200,400 > gpurun_out/r3_run6/ab_c4.txt 2>&1
python tools/layout_ab.py u8 52,100 > gpurun_out/r3_run6/ab_u8.txt 2>&1
python tools/layout_ab.py f32 52,100,200 > gpurun_out/r3_run6/ab_f32.txt 2>&1
cat gpurun_out/r3_run6/ab_*.txt
