#!/bin/bash
# round 4, run 18: more than two lanes for large host batches? (FLATNAV_BIG_LANES: developer knob)
O=gpurun_out/r4_run18; mkdir -p $O
for L in 2 4; do
  echo "== FLATNAV_BIG_LANES=$L" >> $O/threads.txt
  FLATNAV_BIG_LANES=$L timeout 300 python tools/dev/host_threads_bench.py --config c2 --ef 52 >> $O/threads.txt 2>$O/err_$L.txt
  FLATNAV_BIG_LANES=$L timeout 300 python tools/dev/host_threads_bench.py --config c2-uint8 --ef 52 --threads 2,3,4 >> $O/threads.txt 2>>$O/err_$L.txt
done
cat $O/threads.txt
