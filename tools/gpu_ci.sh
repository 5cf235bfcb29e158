#!/bin/bash
# Runs GPU steps in order on the gpurun box; stops the chain when a step was killed by its timeout
# (exit 124/137) -- a plain failure (non-zero exit) is recorded and the next step still runs.
# usage: tools/gpu_ci.sh step1 [step2 ...]   steps: smoke tests tests_full bench prof
mkdir -p gpurun_out
ROOTDIR=$PWD
run() {
  local name=$1; shift
  echo "=== $name: $*"
  "$@" > "$ROOTDIR/gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "=== $name rc=$rc"
  tail -n 12 "$ROOTDIR/gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name was killed; stopping"; exit $rc; fi
  return 0
}
for step in "$@"; do
  case $step in
    smoke) run smoke timeout -k 10 300 python __graft_entry__.py --smoke ;;
    tests) run tests timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "not full_size" ;;
    tests_full) run tests_full timeout -k 10 900 python -m pytest tests -m gpu -x -q ;;
    bench) run bench timeout -k 10 600 python bench.py --steps 10 --warmup 3 ;;
    prof) export TMPDIR=/tmp; R=$PWD; rm -rf gpurun_out/prof; (cd /tmp && run_in() { :; }); \
          run prof timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof" -- python3 "$R/bench.py" --no-cpu-baseline --no-host-api ;;
    pmc_fetch) export TMPDIR=/tmp; R=$PWD; rm -rf gpurun_out/pmc_fetch; \
          run pmc_fetch timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$R/gpurun_out/pmc_fetch" -- python3 "$R/bench.py" --steps 3 --warmup 1 --prewarm-ms 0 --no-cpu-baseline ;;
    pmc_write) export TMPDIR=/tmp; R=$PWD; rm -rf gpurun_out/pmc_write; \
          run pmc_write timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$R/gpurun_out/pmc_write" -- python3 "$R/bench.py" --steps 3 --warmup 1 --prewarm-ms 0 --no-cpu-baseline ;;
    pmc_sq) export TMPDIR=/tmp; R=$PWD; rm -rf gpurun_out/pmc_sq; \
          run pmc_sq timeout -k 10 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$R/gpurun_out/pmc_sq" -- python3 "$R/bench.py" --steps 3 --warmup 1 --prewarm-ms 0 --no-cpu-baseline ;;
    *) echo "unknown step $step" ;;
  esac
done
