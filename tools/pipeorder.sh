#!/bin/bash
# tools/pipeorder.hip: L2-hit probe latency on a CU whose other waves stream from HBM, by the FORM of the stream's requests
cd "${GRAFT_REPO_ROOT:-/root/repo}"
B=build/tools/pipeorder
timeout -k 5 60 $B 0 && timeout -k 5 60 $B 1 && for p in 0 1 2 3; do timeout -k 5 60 $B 2 $p || exit 1; done
