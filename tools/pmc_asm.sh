#!/bin/bash
# instruction counts and wait shares of k_assemble per library variant: bash tools/pmc_asm.sh base ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for NAME in "$@"; do
  OUT=$ROOT/gpurun_out/pmca_$NAME; rm -rf $OUT; mkdir -p $OUT
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/p1 -o p1 -- python3 $ROOT/tools/run_variant.py $NAME --steps 3 > $OUT/p1.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/p2 -o p2 -- python3 $ROOT/tools/run_variant.py $NAME --steps 3 > $OUT/p2.log 2>&1
  echo "== $NAME"; python3 $ROOT/tools/pmc_summary.py $OUT | grep -A18 k_assemble
done
