#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2_run14; mkdir -p $O
FLATNAV_HIP_LIB=$PWD/flatnav_amd/libflatnav_hip_noprefetch.so timeout 900 python tools/r2_probe.py time 1000000 sift > $O/time_sift_noprefetch.txt 2>&1; grep "sorted" $O/time_sift_noprefetch.txt | tail -30
FLATNAV_HIP_LIB=$PWD/flatnav_amd/libflatnav_hip_noprefetch.so timeout 900 python tools/r2_probe.py time 1000000 glove > $O/time_glove_noprefetch.txt 2>&1; grep "sorted" $O/time_glove_noprefetch.txt | tail -30
