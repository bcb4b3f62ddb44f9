#!/bin/bash
# gpurun with a wait for a free GPU slot: retries ONLY the "no box or slot free, nothing charged" answer (exit code 3).
#   tools/gpurun_retry.sh <timeout-seconds> '<command>'
T=$1; shift
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 45
done
exit 3
