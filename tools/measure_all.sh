#!/bin/bash
# One pass over the bench lines of every BASELINE config (1 GPU): tools/measure_all.sh <out-dir>
set -o pipefail
out=${1:-gpurun_out/measure}
mkdir -p "$out"
timeout -k 10 300 python bench.py > "$out/bench.json" 2> "$out/bench.err" && cut -c1-200 "$out/bench.json" &&
timeout -k 10 400 python bench.py --config 3 > "$out/sliding.json" 2> "$out/sliding.err" && cut -c1-200 "$out/sliding.json" &&
timeout -k 10 400 python bench.py --config 4 --train-graph > "$out/train.json" 2> "$out/train.err" && cut -c1-200 "$out/train.json" &&
timeout -k 10 300 python bench.py --config 5 > "$out/swin.json" 2> "$out/swin.err" && cut -c1-200 "$out/swin.json"
