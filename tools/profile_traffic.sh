#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes only (rocprofv3 --pmc, separate passes) around bench.py; usage as profile_pmc.sh
set -u
OUT=${1:-gpurun_out/pmc_t}; shift || true
ARGS=${@:---steps 3 --warmup 2 --settle 0.01 --sectors 360 --no-cpu-baseline --no-end-to-end --no-extras}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p "$OUT"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/$c" -- python3 bench.py $ARGS > "$OUT/$c.log" 2>&1 || echo "pass $c failed"
done
python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
grep -E "^==|FETCH_SIZE|WRITE_SIZE" "$OUT/summary.txt"
