#!/bin/bash
# SQ / LDS counter passes over one kernel of the DiffSwinUNETR step (bench.py --config 5, eager launches).
# usage: tools/pmc_swin.sh <out dir under gpurun_out> <kernel-name substring>
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$1; NAME=${2:-window_attention}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {
  local tag=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$tag --output-format csv -- python3 $ROOT/bench.py --config 5 --steps 3 --warmup 1 --no-graph --no-cpu-baseline > $OUT/$tag.log 2>&1
  python3 $ROOT/tools/pmc_summary.py $OUT/$tag $NAME > $OUT/$tag.json
}
pass A SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
pass B SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
cat $OUT/A.json $OUT/B.json
