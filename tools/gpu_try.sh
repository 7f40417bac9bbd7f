#!/bin/bash
# usage: tools/gpu_try.sh <timeout-s> '<command>'  -- retries while the pod's GPU slots are busy (rc 3)
T=$1; shift
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 60
done
exit 3
