#!/bin/bash
# round 3, run 18: fnv_tune times every candidate from cold caches: is the layout choice stable and right under the bench protocol?
mkdir -p gpurun_out/r3_run18
python tools/layout_ab.py c4 110,200,400 > gpurun_out/r3_run18/ab_c4.txt 2>&1
python tools/layout_ab.py f32 52,100,200 > gpurun_out/r3_run18/ab_f32.txt 2>&1
python tools/layout_ab.py u8 52,100 > gpurun_out/r3_run18/ab_u8.txt 2>&1
grep -v amdgpu gpurun_out/r3_run18/ab_*.txt
