#!/bin/bash
# average duration of the side kernels (gather, layout) per library variant: bash tools/gather_time.sh cur gu4 ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for NAME in "$@"; do
  OUT=$ROOT/gpurun_out/gt_$NAME; rm -rf $OUT; mkdir -p $OUT
  timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $OUT -o r -- python3 $ROOT/tools/run_variant.py $NAME --steps 30 > $OUT/log.txt 2>&1
  python3 $ROOT/tools/rocpd_stats.py $(ls $OUT/*results.db $OUT/*/*results.db 2>/dev/null | head -1) --last 25 | grep -E "assemble|gather|layout|offsets|k_encode" | awk -F, -v n=$NAME '{printf "%-8s %-60s avg %8.1f us\n", n, substr($1,1,60), $4/1000}'
done
