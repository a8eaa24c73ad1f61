#!/bin/bash
# Instruction-fetch side of the encode kernels (is the straight-line, fully unrolled code bound by the front end?):
#   usage (GPU box): bash tools/pmc_ifetch.sh <variant|base> <runs|tiles>
set -u
NAME=$1; PATHSEL=$2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmci_${NAME}_${PATHSEL}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  timeout -k 10 90 rocprofv3 --pmc $group --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- \
      python3 $ROOT/tools/run_variant.py $NAME --steps 3 --path $PATHSEL > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; }
done <<GROUPS
SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_CYCLES SQ_INSTS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_MISC
SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB SQC_TC_INST_REQ SQC_TC_STALL
SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_BUSY_CYCLES
GROUPS
python3 $ROOT/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
rm -rf $OUT/p*/
sed -n '/k_encode/,/k_gather/p' $OUT/summary.txt
