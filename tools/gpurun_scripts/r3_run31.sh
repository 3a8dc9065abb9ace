#!/bin/bash
# round 3, run 31: time per visited-table size at the wide beams (c4 ef=400, c2 ef=200/400), base vs stash + link-row guess
mkdir -p gpurun_out/r3_run31
O=gpurun_out/r3_run31
for cfg in "c4 400" "c2 200,400"; do
  set -- $cfg
  for lib in _base ""; do
    FLATNAV_HIP_LIB=$PWD/flatnav_amd/libflatnav_hip$lib.so timeout 900 python tools/stash_ab.py $1 $2 >> $O/lines.txt 2>$O/err_$1$lib.txt
  done
done
sort -s -k2,2 -k4,4n -k6,6n $O/lines.txt
