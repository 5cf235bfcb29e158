#!/bin/bash
# Runs a list of GPU steps in order on the gpurun box: one "name|timeout_seconds|command" per line of the file given
# as $1.  Every step logs to gpurun_out/<name>.log; a step killed by its timeout (exit 124/137) stops the chain
# (nothing else touches the GPU after a hang); a plain failure is recorded and the next step still runs.
mkdir -p gpurun_out
export TMPDIR=/tmp
while IFS='|' read -r name tmo cmd; do
  [ -z "$name" ] && continue
  case "$name" in \#*) continue ;; esac
  echo "=== $name (limit ${tmo}s): $cmd"
  timeout -k 10 "$tmo" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "=== $name rc=$rc"
  tail -n 15 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name was killed; stopping"; exit $rc; fi
done < "$1"
exit 0
