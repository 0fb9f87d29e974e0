#!/bin/bash
# Diagnostic builds of libdua_hip.so OUT OF the product tree: objects and library go to tools/diag/<name>/, the shipped
# library and its objects are never touched.  Select one at run time with DUA_HIP_LIB=<path to the .so>.
#   tools/build_diag.sh stamp -DDUA_STAMP        in-kernel cycle stamps of the convolution kernel (tools/stamp_conv.py)
# Flags come from the product Makefile; compiler errors are shown.
set -e -o pipefail
name=$1; shift
here=$(cd "$(dirname "$0")" && pwd)
src=$here/../diff_unet_amos_amd/csrc
out=$here/diag/$name
mkdir -p "$out"
flags=$(make -s -C "$src" -pn 2>/dev/null | sed -n 's/^CXXFLAGS = //p' | head -1 | sed 's/\$(ARCH)/gfx950/')
objs=()
for f in "$src"/*.hip; do
  o=$out/$(basename "${f%.hip}").o
  /opt/rocm/bin/hipcc $flags "$@" -c "$f" -o "$o" &
  objs+=("$o")
  while [ "$(jobs -r | wc -l)" -ge 8 ]; do wait -n; done
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$out/libdua_hip.so" "${objs[@]}"
echo "$out/libdua_hip.so"
