set -o pipefail
mkdir -p gpurun_out/r3m
timeout -k 10 200 python bench.py > gpurun_out/r3m/bench.json 2> gpurun_out/r3m/bench.err && cut -c1-200 gpurun_out/r3m/bench.json &&
timeout -k 10 300 python bench.py --dtype f32 > gpurun_out/r3m/bench_f32.json 2> gpurun_out/r3m/bench_f32.err && cut -c1-200 gpurun_out/r3m/bench_f32.json &&
timeout -k 10 400 python bench.py --config 3 > gpurun_out/r3m/sliding.json 2> gpurun_out/r3m/sliding.err && cut -c1-200 gpurun_out/r3m/sliding.json &&
timeout -k 10 400 python bench.py --config 4 --train-graph > gpurun_out/r3m/train.json 2> gpurun_out/r3m/train.err && cut -c1-200 gpurun_out/r3m/train.json &&
timeout -k 10 300 python bench.py --config 5 > gpurun_out/r3m/swin.json 2> gpurun_out/r3m/swin.err && cut -c1-200 gpurun_out/r3m/swin.json
