#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run26; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "sorted_beam_kernel_stays_exact or spill_paths or tail_goes" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for kind in sift sift_u8 glove; do
  for mb in 1 2; do
    timeout 600 python tools/occ_probe.py $kind 32,52,64 merged_beam=$mb 2>&1 | grep -v amdgpu | grep sorted | sed "s/^/mb$mb /" | tee -a $O/occ.txt
  done
done
timeout 600 python tools/latency_probe.py 1000000 merged_beam=2 2>&1 | grep -v amdgpu | grep -v "batch  *[0-9]*[46]:" | tee -a $O/latency.txt
