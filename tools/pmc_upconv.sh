#!/bin/bash
# SQ/LDS counter passes over the folded up-convolution kernel (tools/bench_upconv.py level0).
# usage: tools/pmc_upconv.sh <out dir under gpurun_out> [level0|level1]
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$1; LV=${2:-level0}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {
  local tag=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$tag --output-format csv -- python3 $ROOT/tools/bench_upconv.py $LV 2 > $OUT/$tag.log 2>&1
  python3 $ROOT/tools/pmc_summary.py $OUT/$tag upconv_k3 > $OUT/$tag.json
}
pass A SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
pass B SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
pass C FETCH_SIZE
pass D WRITE_SIZE
cat $OUT/A.json $OUT/B.json $OUT/C.json $OUT/D.json
