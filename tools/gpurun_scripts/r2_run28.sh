#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run28; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "sorted_beam_kernel_stays_exact or spill_paths or tail_goes" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for kind in glove sift; do
  for mb in 0 1; do
    timeout 600 python tools/occ_probe.py $kind 300,400,800 merged_beam=$mb 2>&1 | grep -v amdgpu | grep sorted | sed "s/^/mb$mb /" | tee -a $O/occ.txt
  done
done
