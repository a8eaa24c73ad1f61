#!/bin/bash
# Round-4 PMC passes (the round-3 recipe; the summary now also lists k_assemble) over ONE library variant and path: instruction mix, what the waves wait for, LDS, and the memory path
# (texture addresser, L1, L2, fabric).  One rocprofv3 run per counter group, --kernel-trace only beside --pmc.
#   usage (GPU box): bash tools/pmc_r04.sh <variant|base> <runs|tiles> [W H N]
set -u
NAME=$1; PATHSEL=$2; W=${3:-1920}; H=${4:-1080}; N=${5:-300}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc4_${NAME}_${PATHSEL}_${W}x${H}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  timeout -k 10 90 rocprofv3 --pmc $group --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- \
      python3 $ROOT/tools/run_variant.py $NAME --steps 3 --path $PATHSEL --w $W --h $H --n $N > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; }
done <<GROUPS
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE
FETCH_SIZE
WRITE_SIZE
GRBM_GUI_ACTIVE TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum
TCC_REQ_sum TCC_READ_sum TCC_BUSY_sum TCC_TAG_STALL_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_HIT_sum TCC_MISS_sum
GROUPS
python3 $ROOT/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
rm -rf $OUT/p*/   # the raw CSV trees are large; the summary (with the mean dispatch durations) is what gets committed
cat $OUT/summary.txt
