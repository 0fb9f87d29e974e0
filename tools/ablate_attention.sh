#!/bin/bash
# Ablation builds of the windowed-attention kernel (-DWA_ABL=bits: 1 no bias-table lookup, 2 no exp2, 4 no shift-mask compare,
# 8 no running maximum / rescale, 16 staging only), each timed on the stage-0 launch of bench.py --config 5 under rocprofv3.
# Results are WRONG by construction, so every variant is built OUT of the product tree (tools/build_diag.sh ->
# tools/diag/wa_abl_<bits>/libdua_hip.so, flags taken from the product Makefile, compiler errors shown) and selected with
# DUA_HIP_LIB: the shipped library and its objects are never touched, an interrupted run leaves nothing to restore.
# usage (on the GPU box): tools/ablate_attention.sh <out dir>
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/$1; mkdir -p "$OUT"
for v in ${ABL_LIST:-0 1 2 4 8 15 16}; do
  lib=$("$ROOT/tools/build_diag.sh" wa_abl_$v -DWA_ABL=$v | tail -1)
  export DUA_HIP_LIB=$lib
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace -d "$OUT/p$v" -- python3 "$ROOT/bench.py" --config 5 --steps 10 --warmup 2 --no-roofline --no-cpu-baseline > "$OUT/p$v.log" 2>&1)
  unset DUA_HIP_LIB
  echo "WA_ABL=$v"; python3 "$ROOT/tools/rocpd_stats.py" "$OUT"/p$v/*/*_results.db --by-grid window_attention | grep grid | cut -c60-200
  rm -rf "$OUT/p$v"
done
