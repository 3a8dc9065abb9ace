#!/bin/bash
# round 4, run 2: parity of the query-in-registers tree; 768-d variants (waves per SIMD x passes in flight, table sizes, and the
# timing-only no-bitmap build) A/B'ed in one process on one copy of the 10M x 768 index; c4 / c2 regression A/B against the r3 library
O=gpurun_out/r4_run2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_configs.py -m gpu -x -q -k "not fullsize" > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
E=flatnav_amd/_exp
timeout 1500 python tools/dev/knob_sweep.py --config c3-lowrank --ef 800,700 --rounds 2 --steps 5 \
  --libs r3=$E/libflatnav_hip_r3.so,w3p4=$E/lib768_w3p4.so,w3p6=$E/lib768_w3p6.so,w2p8=$E/lib768_w2p8.so,nobm=$E/lib768_nobm.so,w2p8nobm=$E/lib768_w2p8nobm.so \
  --sets base r3:base w3p4:base w3p6:base w2p8:base w2p8:visited_slots=16384 w2p8:visited_slots=8192 w3p6:visited_slots=8192 \
         visited_slots=8192 visited_slots=1536 nobm:base w2p8nobm:base > $O/c3.txt 2>$O/c3.err
cat $O/c3.txt
timeout 400 python tools/dev/knob_sweep.py --config c4 --ef 110,200,400 --rounds 3 --steps 10 --libs r3=$E/libflatnav_hip_r3.so --sets base r3:base > $O/c4.txt 2>$O/c4.err
cat $O/c4.txt
timeout 400 python tools/dev/knob_sweep.py --config c2 --ef 52,100 --rounds 3 --steps 20 --libs r3=$E/libflatnav_hip_r3.so --sets base r3:base > $O/c2.txt 2>$O/c2.err
cat $O/c2.txt
