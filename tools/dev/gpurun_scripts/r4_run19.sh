#!/bin/bash
# round 4, run 19: four big lanes where the workspace is small -- tests, the c2 / c2-uint8 / c4 lines with their host rates, and a
# 10M-node configuration (two lanes only)
O=gpurun_out/r4_run19; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_python_api.py tests/test_gpu_multi_device.py tests/test_gpu_bench.py -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
python bench.py --secondary-configs c2-uint8,c4,c3 --steps 20 > $O/bench.json 2> $O/bench.err
grep "host-buffer" $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); print('\n'.join(d['summary']))"
