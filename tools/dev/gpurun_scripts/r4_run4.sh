#!/bin/bash
# round 4, run 4: the suite's kernel / round-4 / multi-device files on the tree with tail shadows (variant 6) and the
# host pipeline; then every variant pinned, A/B in one process per configuration; then the host path timed properly
O=gpurun_out/r4_run4; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py tests/test_gpu_parity.py tests/test_gpu_multi_device.py -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -12 $O/pytest.log
V="sorted_variant=1 sorted_variant=6 sorted_variant=2 sorted_variant=3 sorted_variant=4 sorted_variant=5 sorted_variant=0"
timeout 300 python tools/dev/knob_sweep.py --config c2 --ef 52,100 --rounds 3 --steps 20 --no-tune --sets $V > $O/c2.txt 2>$O/c2.err; cat $O/c2.txt
timeout 300 python tools/dev/knob_sweep.py --config c2-uint8 --dtype uint8 --ef 52 --rounds 3 --steps 20 --no-tune --sets $V > $O/u8.txt 2>$O/u8.err; cat $O/u8.txt
timeout 300 python tools/dev/knob_sweep.py --config c4 --ef 110,200 --rounds 3 --steps 10 --no-tune --sets $V > $O/c4.txt 2>$O/c4.err; cat $O/c4.txt
timeout 600 python tools/dev/knob_sweep.py --config c5-lowrank --ef 80 --rounds 3 --steps 10 --no-tune --sets $V > $O/c5l.txt 2>$O/c5l.err; cat $O/c5l.txt
timeout 600 python tools/dev/knob_sweep.py --config c3-lowrank --ef 700 --rounds 2 --steps 4 --no-tune --sets sorted_variant=1 sorted_variant=6 sorted_variant=3 sorted_variant=2 > $O/c3.txt 2>$O/c3.err; cat $O/c3.txt
timeout 300 python tools/dev/host_path_bench.py --config c2 --ef 52 > $O/host_c2.txt 2>$O/host_c2.err; cat $O/host_c2.txt
