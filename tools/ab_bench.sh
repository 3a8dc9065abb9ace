#!/bin/bash
# Developer tool: run bench.py against several library builds / option sets and print one line each.
# usage: tools/ab_bench.sh "<label>|<lib path or empty>|<extra bench args>" ...
for spec in "$@"; do
  IFS='|' read -r label lib args <<< "$spec"
  if [ -n "$lib" ]; then export FLATNAV_HIP_LIB="$lib"; else unset FLATNAV_HIP_LIB; fi
  python bench.py --no-cpu-baseline $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$label', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['frac'],3), d['config']['recall_at_10'], d['config']['launch'])"
done
