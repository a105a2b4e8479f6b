#!/bin/bash
# usage: bench_compare.sh VAR v1 v2 ...  -> ms/step and cells/s of bench.py for each value of env VAR
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v timeout 300 python bench.py --no-cpu-baseline --no-fused 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$VAR=$v', round(d['ms_per_step'], 4), 'ms/step', round(d['value'] / 1e9, 3), 'Gcells/s', 'grad_kernel_ms', round(d['roofline']['kernel_avg_ms'], 4))"
done
