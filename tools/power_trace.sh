#!/bin/bash
# Samples socket power and shader clock (rocm-smi) while bench.py runs a long sustained pass of one path:
#   usage (GPU box): bash tools/power_trace.sh <runs|tiles> [steps]
P=${1:-runs}; STEPS=${2:-6000}
( for i in $(seq 1 40); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Graphics Package Power" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.15; done ) > /tmp/power_$P.txt &
SAMPLER=$!
python3 bench.py --no-cpu-baseline --deliver none --path $P --steps $STEPS --warmup 50 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$P', 'fps', round(d['value']), 'ms_per_step', d['ms_per_step'], 'kernel_ms', r.get('kernel_ms'), 'frac', r['frac'])"
wait $SAMPLER
cat /tmp/power_$P.txt
