#!/bin/bash
# round 3, run 26: the overflow stash behind the visited tag table -- parity tests first, then time per table size, base vs stash
mkdir -p gpurun_out/r3_run26
O=gpurun_out/r3_run26
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_configs.py -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for cfg in "c2 52,100" "u8 52" "c4 110,200"; do
  set -- $cfg
  for lib in _base ""; do
    FLATNAV_HIP_LIB=$PWD/flatnav_amd/libflatnav_hip$lib.so timeout 600 python tools/stash_ab.py $1 $2 >> $O/lines.txt 2>$O/err_$1$lib.txt
  done
done
cat $O/lines.txt
