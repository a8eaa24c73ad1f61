#!/bin/bash
# HBM traffic of the encode kernel (FETCH_SIZE / WRITE_SIZE, one rocprofv3 pass each) per library variant:
#   bash tools/pmc_traffic.sh W H N name...      prints (2 x FETCH_SIZE + WRITE_SIZE) KiB as bytes and the ratio to 3*W*H*N + output
W=$1; H=$2; N=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for NAME in "$@"; do
  OUT=$ROOT/gpurun_out/pmct_${NAME}_${W}x${H}; rm -rf $OUT; mkdir -p $OUT
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/p_$c -o p -- python3 $ROOT/tools/run_variant.py $NAME --steps 3 --w $W --h $H --n $N > $OUT/$c.log 2>&1
  done
  python3 $ROOT/tools/pmc_summary.py $OUT | awk -v n=$NAME -v px=$((W*H*3*N)) '
    /^k_/ {k=$1} /FETCH_SIZE/ {sub(/.*mean=/,""); f[k]=$0} /WRITE_SIZE/ {sub(/.*mean=/,""); w[k]=$0} /DURATION/ {sub(/.*mean=/,""); d[k]=$0}
    END {for (k in f) printf "%-8s %-16s fetch %.0f KiB write %.0f KiB -> %.0f bytes = %.4f x the pixel bytes, %.1f us under the profiler\n", n, k, f[k], w[k], (2*f[k]+w[k])*1024, (2*f[k]+w[k])*1024/px, d[k]/1000}'
done
