#!/bin/bash
# memory-path PMC passes (texture addresser, L1, L2) over a library variant: bash tools/pmc_mempath.sh <variant>
set -u
NAME=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcm_$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  timeout -k 10 60 rocprofv3 --pmc $group --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- \
      python3 $ROOT/tools/run_variant.py $NAME --steps 3 > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; }
done <<GROUPS
GRBM_GUI_ACTIVE TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TCP_GATE_EN1_sum TCP_GATE_EN2_sum
TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum
TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum
TCC_REQ_sum TCC_READ_sum TCC_BUSY_sum TCC_TAG_STALL_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_HIT_sum TCC_MISS_sum
GROUPS
python3 $ROOT/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
grep -A40 k_encode_dense $OUT/summary.txt | head -40
