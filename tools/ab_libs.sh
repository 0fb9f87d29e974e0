#!/bin/bash
# The bench lines of configs 2, 5 and 4 with the shipped library and with ANOTHER build of it (tools/build_diag.sh), alternating
# on one box: tools/ab_libs.sh <other libdua_hip.so> <out file> [rounds]
alt=$(realpath "$1"); out=${2:-gpurun_out/ab_libs.txt}; rounds=${3:-2}
: > "$out"
run() {  # label, env prefix, bench args
  local ms
  ms=$(env $2 python bench.py $3 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "import sys, json; print(round(json.loads(sys.stdin.readline())['ms_per_step'], 4))")
  echo "$1 $3: $ms ms" | tee -a "$out"
}
for r in $(seq "$rounds"); do
  for cfg in "--config 2 --no-full-loop --no-f32 --steps 300" "--config 5 --steps 200" "--config 4 --train-graph"; do
    run shipped "DUA_X=0" "$cfg"
    run other "DUA_DEBUG=1 DUA_HIP_LIB=$alt" "$cfg"
  done
done
