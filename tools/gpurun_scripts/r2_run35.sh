#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run35; mkdir -p $O
FNV_FULLSIZE=1 timeout 3000 python -m pytest tests/test_gpu_configs.py -q -k "fullsize" > $O/pytest_fullsize.log 2>&1; tail -6 $O/pytest_fullsize.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -3
