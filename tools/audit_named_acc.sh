#!/bin/bash
# ISA audit of a translation unit whose kernels keep MFMA accumulators in v[128:255] by name (csrc/named_acc.hpp):
# outside ;;#ASMSTART / ;;#ASMEND no instruction may name v128..v255 (or an AGPR) -- the compiler believes those registers
# do not exist -- and the kernels listed must not use scratch when asked (--no-scratch).
#   tools/audit_named_acc.sh <file.hip> [--no-scratch] [extra hipcc flags]
set -e -o pipefail
src=$1; shift
noscratch=0
if [ "$1" = "--no-scratch" ]; then noscratch=1; shift; fi
tmp=$(mktemp -d)
trap 'rm -rf "$tmp"' EXIT
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form \
  "$@" -S --cuda-device-only "$src" -o "$tmp/k.s" 2>/dev/null
awk '
  /;;#ASMSTART/ { inasm = 1; next }
  /;;#ASMEND/   { inasm = 0; next }
  /^[_A-Za-z0-9.$]+:/ { label = $1 }
  !inasm && /^\t[a-z]/ {
    line = $0
    sub(/;.*/, "", line)
    if (line ~ /[^a-z_0-9]a\[?[0-9]/ || line ~ /v_accvgpr/) { print "AGPR outside asm (" label "): " $0; bad = 1 }
    # single registers v128..v255 and ranges v[lo:hi] with hi >= 128
    n = split(line, tok, /[ ,\t]+/)
    for (i = 1; i <= n; ++i) {
      t = tok[i]
      if (t ~ /^v[0-9]+$/) { r = substr(t, 2) + 0; if (r >= 128) { print "accumulator register outside asm (" label "): " $0; bad = 1 } }
      else if (t ~ /^v\[[0-9]+:[0-9]+\]$/) { split(substr(t, 3, length(t) - 3), ab, ":"); if (ab[2] + 0 >= 128) { print "accumulator register outside asm (" label "): " $0; bad = 1 } }
    }
  }
  END { exit bad }
' "$tmp/k.s"
if [ $noscratch = 1 ] && grep -q "scratch_" "$tmp/k.s"; then
  echo "scratch accesses:"; grep -c "scratch_" "$tmp/k.s"; exit 1
fi
echo "audit ok: $(grep -c 'v_mfma' "$tmp/k.s") MFMA statements, $(grep -c 'scratch_' "$tmp/k.s" || true) scratch accesses"
