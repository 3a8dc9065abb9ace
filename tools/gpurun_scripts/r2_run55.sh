#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run55; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "sorted_beam or random_shapes" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for kind in sift glove; do
    timeout 600 python tools/occ_probe.py $kind 65,100,128 2>&1 | grep -v amdgpu | grep sorted | tee -a $O/occ.txt
done
timeout 600 python tools/latency_probe.py 1000000 2>&1 | grep -v amdgpu | grep "search_single" | tee -a $O/latency.txt
