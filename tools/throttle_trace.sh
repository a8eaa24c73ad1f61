#!/bin/bash
# Full power-management snapshot (amd-smi / rocm-smi) in the middle of a long sustained pass of one path.
P=${1:-runs}
( sleep 2.0; amd-smi metric -g 0 2>&1 | head -150 > /tmp/amdsmi_$P.txt; rocm-smi -a 2>/dev/null | grep -iE "temp|volt|power|clk|throttl|perf|fan" | head -40 > /tmp/rocmsmi_$P.txt ) &
S=$!
python3 bench.py --no-cpu-baseline --deliver none --path $P --steps 9000 --warmup 50 > /dev/null 2>&1
wait $S
echo "=== $P amd-smi"; cat /tmp/amdsmi_$P.txt; echo "=== $P rocm-smi"; cat /tmp/rocmsmi_$P.txt
