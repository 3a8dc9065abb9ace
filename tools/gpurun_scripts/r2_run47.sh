#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run47; mkdir -p $O
timeout 1500 python tools/bigid_check.py 20000000 2>&1 | grep -v amdgpu | tee $O/bigid.txt
