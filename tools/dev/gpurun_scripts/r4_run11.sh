#!/bin/bash
# round 4, run 11: the guessed row's bitmap words one hop ahead -- parity (forced tiny tables take the new path on most hops),
# then A/B against the library without it (r4a) on the wide-beam and the headline configurations
O=gpurun_out/r4_run11; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_configs.py -m gpu -x -q -k "not fullsize" > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
E=flatnav_amd/_exp
timeout 400 python tools/dev/knob_sweep.py --config c4 --ef 110,200,400 --rounds 3 --steps 10 --libs r4a=$E/libflatnav_hip_r4a.so --sets base r4a:base > $O/c4.txt 2>$O/c4.err; cat $O/c4.txt
timeout 400 python tools/dev/knob_sweep.py --config c2 --ef 52,100,200,400 --rounds 3 --steps 10 --libs r4a=$E/libflatnav_hip_r4a.so --sets base r4a:base > $O/c2.txt 2>$O/c2.err; cat $O/c2.txt
timeout 900 python tools/dev/knob_sweep.py --config c3-lowrank --ef 700 --rounds 3 --steps 4 --libs r4a=$E/libflatnav_hip_r4a.so --sets base r4a:base > $O/c3.txt 2>$O/c3.err; cat $O/c3.txt
timeout 900 python tools/dev/knob_sweep.py --config c3 --ef 200 --rounds 3 --steps 6 --libs r4a=$E/libflatnav_hip_r4a.so --sets base r4a:base > $O/c3w.txt 2>$O/c3w.err; cat $O/c3w.txt
