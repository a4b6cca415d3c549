#!/bin/bash
# gpurun, waiting for a free slot: retries ONLY when gpurun reports exit code 3 (no box or slot, nothing ran, nothing charged)
# usage: tools/gpurun_wait.sh <timeout-seconds> '<command>'
t=$1; shift
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
