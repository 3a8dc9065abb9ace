#!/bin/bash
# round 4, run 26: the granule-aware grid and heap trim (this tree) against the library before it (old), alternating on one
# copy of each index; then parity where launch geometry matters
O=gpurun_out/r4_run26; mkdir -p $O
E=flatnav_amd/_exp
timeout 300 python tools/dev/knob_sweep.py --config c2 --dtype uint8 --ef 52,100 --rounds 3 --steps 10 --libs old=$E/libflatnav_hip_old.so --sets base old:base > $O/c2u8.txt 2>$O/c2u8.err; cat $O/c2u8.txt
timeout 300 python tools/dev/knob_sweep.py --config c2 --ef 52,100,200 --rounds 3 --steps 10 --libs old=$E/libflatnav_hip_old.so --sets base old:base > $O/c2.txt 2>$O/c2.err; cat $O/c2.txt
timeout 300 python tools/dev/knob_sweep.py --config c4 --ef 110,200,400 --rounds 3 --steps 8 --libs old=$E/libflatnav_hip_old.so --sets base old:base cand_slots=314 > $O/c4.txt 2>$O/c4.err; cat $O/c4.txt
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_round4.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
