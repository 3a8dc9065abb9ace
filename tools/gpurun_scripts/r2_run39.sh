#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run39; mkdir -p $O
timeout 900 python tools/fuzz_debug.py $O 2>&1 | grep -v amdgpu | tee $O/fuzz.txt
ls -la $O
