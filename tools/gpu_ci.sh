#!/bin/bash
# Runs GPU steps in order on the gpurun box; stops the chain when a step was killed by its timeout
# (exit 124/137) -- a plain failure (non-zero exit) is recorded and the next step still runs.
# usage: tools/gpu_ci.sh step1 [step2 ...]   steps: smoke tests tests_full bench prof
mkdir -p gpurun_out
run() {
  local name=$1; shift
  echo "=== $name: $*"
  "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "=== $name rc=$rc"
  tail -n 12 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name was killed; stopping"; exit $rc; fi
  return 0
}
for step in "$@"; do
  case $step in
    smoke) run smoke timeout -k 10 300 python __graft_entry__.py --smoke ;;
    tests) run tests timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "not full_size" ;;
    tests_full) run tests_full timeout -k 10 900 python -m pytest tests -m gpu -x -q ;;
    bench) run bench timeout -k 10 600 python bench.py --steps 10 --warmup 3 ;;
    *) echo "unknown step $step" ;;
  esac
done
