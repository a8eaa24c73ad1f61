#!/bin/bash
# instruction-count PMC passes over a library variant: bash tools/pmc_variant.sh <variant> [tag]
set -u
NAME=$1; TAG=${2:-$1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $group --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- \
      python3 $ROOT/tools/run_variant.py $NAME --steps 3 > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; }
done <<GROUPS
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH
SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 GRBM_GUI_ACTIVE
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_IFETCH_LEVEL
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum
FETCH_SIZE
SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_SMEM SQ_BUSY_CU_CYCLES
GROUPS
python3 $ROOT/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
grep -A40 k_encode_dense $OUT/summary.txt | head -45
