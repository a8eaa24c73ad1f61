#!/bin/bash
# Alternates `bench.py --path runs` and `--path tiles` (separate processes, the way the driver runs the bench) N times on one box
# and prints step and kernel times: the in-process A/B (tools/ab.py) times bursts of five launches, this times sustained runs.
#   usage (GPU box): bash tools/bench_paths.sh [N] [extra bench args...]
N=${1:-3}; shift
for i in $(seq 1 $N); do
  for p in runs tiles; do
    timeout -k 10 120 python3 bench.py --no-cpu-baseline --deliver none --path $p "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$p', 'fps', round(d['value']), 'ms_per_step', d['ms_per_step'], 'kernel_ms', r.get('kernel_ms'), 'frac', r['frac'])"
  done
done
