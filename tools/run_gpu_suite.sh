#!/bin/bash
# GPU test suite in ONE process, with evidence kept if the interpreter dies: Python's faulthandler dumps every thread's
# Python stack, a core file (if the box writes one next to the process) is read back with gdb for the native stacks.
#   tools/run_gpu_suite.sh <out-dir> [extra pytest arguments]
out=${1:-gpurun_out/suite}; shift
mkdir -p "$out"
ulimit -c unlimited 2>/dev/null
export PYTHONFAULTHANDLER=1
timeout -k 10 1100 python -X faulthandler -m pytest tests -m gpu -x -q "$@" > "$out/tests.log" 2>&1
rc=$?
tail -n 5 "$out/tests.log"
if [ $rc -ge 128 ] || grep -q "Fatal Python error" "$out/tests.log"; then
  echo "interpreter died (rc=$rc): looking for a core file" | tee -a "$out/tests.log"
  core=$(ls -t core core.* /tmp/core* 2>/dev/null | head -n 1)
  if [ -n "$core" ] && command -v gdb >/dev/null; then
    gdb -batch -ex "thread apply all bt 40" "$(command -v python3)" "$core" > "$out/core_backtrace.txt" 2>&1
    tail -n 80 "$out/core_backtrace.txt"
  else
    echo "no core file / no gdb" | tee -a "$out/tests.log"
  fi
fi
exit $rc
