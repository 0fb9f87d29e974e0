#!/bin/bash
# SQ/LDS counter passes over the conv3d_k3 kernel on the big layer shapes (tools/bench_conv.py).
# usage: tools/pmc_conv.sh <out dir under gpurun_out> <variants> <shape indices> [kernel-name substring]
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$1; VAR=${2:-0}; IDX=${3:-1,2}; NAME=${4:-conv3d_k3}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {
  local tag=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$tag --output-format csv -- python3 $ROOT/tools/bench_conv.py $VAR $IDX 2 > $OUT/$tag.log 2>&1
  python3 $ROOT/tools/pmc_summary.py $OUT/$tag $NAME > $OUT/$tag.json
}
pass A SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
pass B SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
pass C SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE
cat $OUT/A.json $OUT/B.json $OUT/C.json
