#!/bin/bash
# rocprofv3 --kernel-trace --stats over the DEFAULT bench command (the summary the bench line's roofline.avg_launch_ms must agree
# with), plus the kernel timeline of one replayed step: tools/profile_bench.sh <out-dir> [bench.py arguments]
out=$(realpath -m "${1:-gpurun_out/profile}"); shift
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$out/prof" --output-format csv -- python3 "$root/bench.py" --no-cpu-baseline --no-traffic "$@" > "$out/bench_profiled.json" 2> "$out/prof.log"
cd "$root"
python tools/step_timeline.py "$out/prof" > "$out/step_timeline.txt" 2>> "$out/prof.log"
f=$(ls "$out"/prof/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" "$out/kernel_stats.csv"
rm -rf "$out/prof"
tail -n 14 "$out/step_timeline.txt"
