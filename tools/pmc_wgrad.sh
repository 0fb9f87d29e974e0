#!/bin/bash
# SQ / LDS counter passes over the 3x3x3 weight-gradient kernel on its 96^3 layers (tools/bench_wgrad.py --only 2).
# usage: tools/pmc_wgrad.sh <out dir under gpurun_out> [kernel-name substring; default conv3d_k3_wgrad]
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$1
NAME=${2:-conv3d_k3_wgrad}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {
  local tag=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$tag --output-format csv -- python3 $ROOT/tools/bench_wgrad.py --only 2 > $OUT/$tag.log 2>&1
  python3 $ROOT/tools/pmc_summary.py $OUT/$tag $NAME > $OUT/$tag.json
}
pass A SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
pass B SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
pass C SQ_BUSY_CYCLES SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL
cat $OUT/A.json $OUT/B.json $OUT/C.json
