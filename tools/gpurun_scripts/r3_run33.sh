#!/bin/bash
# round 3, run 33: rules' layout vs fnv_tune's layout under the bench protocol, with the stash
mkdir -p gpurun_out/r3_run33
O=gpurun_out/r3_run33
timeout 900 python tools/layout_ab.py c2 52,100,200,400 >> $O/lines.txt 2>$O/err_c2.txt
timeout 900 python tools/layout_ab.py c4 110,200,400 >> $O/lines.txt 2>$O/err_c4.txt
timeout 900 python tools/layout_ab.py u8 52,100 >> $O/lines.txt 2>$O/err_u8.txt
cat $O/lines.txt
