#!/bin/bash
# usage (on the GPU box, from the repo root): tools/session.sh NAME 'cmd1' 'cmd2' ...  -- runs the commands in order, each
# under its own timeout, logging to gpurun_out/NAME/step_K.log; stops at the first one that fails or times out
name=$1; shift
out=gpurun_out/$name
mkdir -p $out
k=0
for c in "$@"; do
  k=$((k+1))
  echo "== step $k: $c" | tee -a $out/steps.log
  timeout -k 10 420 bash -c "$c" > $out/step_$k.log 2>&1
  rc=$?
  echo "   rc=$rc" | tee -a $out/steps.log
  tail -n 12 $out/step_$k.log
  if [ $rc -ne 0 ]; then exit $rc; fi
done
