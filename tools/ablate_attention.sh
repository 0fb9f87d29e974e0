#!/bin/bash
# Ablation builds of the windowed-attention kernel (-DWA_ABL=bits: 1 no bias-table lookup, 2 no exp2, 4 no shift-mask compare,
# 8 no running maximum / rescale, 16 staging only), each timed on the stage-0 launch of bench.py --config 5 under rocprofv3.  Results are WRONG
# by construction; the script restores the normal build at the end.  usage (on the GPU box): tools/ablate_attention.sh <out dir>
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/$1; mkdir -p $OUT
CS=$ROOT/diff_unet_amos_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off"
for v in ${ABL_LIST:-0 1 2 4 8 15 16}; do
  (cd $CS && /opt/rocm/bin/hipcc $FLAGS -DWA_ABL=$v -c window_attention.hip -o build/window_attention.o 2> /dev/null && make > /dev/null)
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace -d $OUT/p$v -- python3 $ROOT/bench.py --config 5 --steps 10 --warmup 2 --no-roofline --no-cpu-baseline > $OUT/p$v.log 2>&1)
  echo "WA_ABL=$v"; python3 $ROOT/tools/rocpd_stats.py $OUT/p$v/*/*_results.db --by-grid window_attention | grep grid | cut -c60-200
  rm -rf $OUT/p$v
done
(cd $CS && touch window_attention.hip && make > /dev/null)
