#!/bin/bash
# ISA audit of a translation unit whose kernels keep MFMA accumulators in v[128:255] by name (csrc/named_acc.hpp).
# For every kernel of the file that contains such statements (an MFMA on v[128..] between ;;#ASMSTART / ;;#ASMEND): outside the
# asm statements no instruction may name v128..v255 or an AGPR -- the compiler believes those registers do not exist -- and,
# with --no-scratch, the kernel must not touch scratch.  Other kernels of the file are not looked at.
#   tools/audit_named_acc.sh <file.hip> [--no-scratch] [extra hipcc flags]
set -e -o pipefail
src=$1; shift
noscratch=0
if [ "$1" = "--no-scratch" ]; then noscratch=1; shift; fi
tmp=$(mktemp -d)
trap 'rm -rf "$tmp"' EXIT
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form \
  "$@" -S --cuda-device-only "$src" -o "$tmp/k.s" 2>/dev/null
awk -v noscratch=$noscratch '
  function flush() {
    if (fn != "" && named) {
      kernels++
      if (nbad) { printf "%s", badtext; bad = 1 }
      if (noscratch && scratch) { print "scratch accesses in " fn ": " scratch; bad = 1 }
      summary = summary sprintf("  %s: %d MFMA statements, %d scratch accesses\n", fn, mfma, scratch)
    }
    named = 0; nbad = 0; badtext = ""; mfma = 0; scratch = 0
  }
  /^_Z[A-Za-z0-9_]+:/ { flush(); fn = $1; sub(/:$/, "", fn) }
  /;;#ASMSTART/ { inasm = 1; next }
  /;;#ASMEND/   { inasm = 0; next }
  inasm && /v_mfma/ { mfma++; if ($0 ~ /v\[(1[2-9][0-9]|2[0-9][0-9]):/) named = 1 }
  /scratch_/ { scratch++ }
  !inasm && /^\t[a-z]/ {
    line = $0
    sub(/;.*/, "", line)
    if (line ~ /[^a-z_0-9]a\[?[0-9]/ || line ~ /v_accvgpr/) { badtext = badtext "AGPR outside asm (" fn "): " $0 "\n"; nbad++ }
    n = split(line, tok, /[ ,\t]+/)
    for (i = 1; i <= n; ++i) {
      t = tok[i]
      if (t ~ /^v[0-9]+$/) { r = substr(t, 2) + 0; if (r >= 128) { badtext = badtext "accumulator register outside asm (" fn "): " $0 "\n"; nbad++ } }
      else if (t ~ /^v\[[0-9]+:[0-9]+\]$/) { split(substr(t, 3, length(t) - 3), ab, ":"); if (ab[2] + 0 >= 128) { badtext = badtext "accumulator register outside asm (" fn "): " $0 "\n"; nbad++ } }
    }
  }
  END { flush(); if (!kernels) { print "no kernel with named accumulators found"; exit 1 }; printf "audit %s: %d kernel(s)\n%s", (bad ? "FAILED" : "ok"), kernels, summary; exit bad }
' "$tmp/k.s"
