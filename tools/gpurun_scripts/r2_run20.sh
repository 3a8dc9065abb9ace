#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run20; mkdir -p $O
for w in "" _w5 _w6 _w8; do
  for kind in sift sift_u8 glove; do
    FLATNAV_HIP_LIB=$PWD/flatnav_amd/libflatnav_hip$w.so timeout 600 python tools/occ_probe.py $kind 32,52,100,200,400 2>&1 | grep -v amdgpu | grep sorted | sed "s/^/lib$w /" | tee -a $O/occ.txt
  done
done
