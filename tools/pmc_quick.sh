#!/bin/bash
# quick instruction-count pass over variants
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for NAME in "$@"; do
  OUT=$ROOT/gpurun_out/pmcq_$NAME; mkdir -p $OUT
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/p1 -o p1 -- python3 $ROOT/tools/run_variant.py $NAME --steps 3 > $OUT/p1.log 2>&1
  python3 $ROOT/tools/pmc_summary.py $OUT | grep -A8 k_encode_dense
done
