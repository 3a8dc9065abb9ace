#!/bin/bash
# Developer tool: time the device build (device search + device wiring) at 1M.
timeout 300 python tools/device_build.py --n 1000000 --max-batch ${1:-65536} --skip-host --wiring device --efs 50,100 2>&1 | grep -v "Warning\|amdgpu.ids"
