#!/bin/bash
# round 4, run 6: the gated host path (tests with the gate on by default, then its timing against the plain path); 768-d
# with two waves per SIMD, eight vectors in flight, a table that holds the whole visited set and the exact search's heap in LDS
O=gpurun_out/r4_run6; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py tests/test_gpu_parity.py tests/test_gpu_multi_device.py tests/test_gpu_python_api.py -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
timeout 300 python tools/dev/host_path_bench.py --config c2 --ef 52 > $O/host_c2.txt 2>$O/host_c2.err; cat $O/host_c2.txt
timeout 300 python tools/dev/host_path_bench.py --config c4 --ef 110 > $O/host_c4.txt 2>$O/host_c4.err; cat $O/host_c4.txt
E=flatnav_amd/_exp
timeout 1200 python tools/dev/knob_sweep.py --config c3-lowrank --ef 700 --rounds 2 --steps 5 \
  --libs w3p6=$E/lib768_w3p6.so,w2p8=$E/lib768_w2p8.so \
  --sets base w2p8:visited_slots=16384 w2p8:visited_slots=8192,sorted_cand_lds=1 w2p8:visited_slots=12288,sorted_cand_lds=1 \
         w2p8:visited_slots=16384,sorted_cand_lds=1 w3p6:visited_slots=8192,sorted_cand_lds=1 w3p6:visited_slots=4096,sorted_cand_lds=1 \
         visited_slots=4096,sorted_cand_lds=1 w2p8:visited_slots=8192 > $O/c3.txt 2>$O/c3.err
cat $O/c3.txt
