#!/bin/bash
# tools/pipeorder.hip: latency of an L2-hit access on a CU whose other waves stream from HBM -- by the form of the stream's
# requests, by the kind of the probe (register load / LDS-direct load / store) and by whether streamers and probers share SIMDs
cd "${GRAFT_REPO_ROOT:-/root/repo}"
B=build/tools/pipeorder
run() { timeout -k 5 60 $B "$@" || exit 1; }
for probe in 0 1 2; do for split in 0 1; do run 0 0 $probe $split; run 1 0 $probe $split; done; done
for p in 0 1 2 3; do run 2 $p 0 0; done
echo "== the CUs take turns (split 2): streams and probes share the L2s and the fabric, not a CU"
for probe in 0 2; do run 0 0 $probe 2; run 1 0 $probe 2; done
