#!/bin/bash
# round 4, run 5: the kernel / round-4 / multi-device test files; the host pipeline's timeline
O=gpurun_out/r4_run5; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py tests/test_gpu_parity.py tests/test_gpu_multi_device.py -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -25 $O/pytest.log
FLATNAV_PIPE_LOG=1 timeout 300 python tools/dev/host_path_bench.py --config c2 --ef 52 --calls 6 > $O/host_c2.txt 2>$O/host_c2.err; cat $O/host_c2.txt; grep "fnv pipeline" $O/host_c2.err | tail -8
