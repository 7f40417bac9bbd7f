#!/bin/bash
# one rocprofv3 --pmc pass around bench.py for an experimental build: tools/pmc_one.sh OUTDIR "COUNTER ..." [bench args]
# (WRP_LIB_PATH selects the library; never combined with tracing other than --kernel-trace)
set -u
OUT=$1; CTRS=$2; shift 2
ARGS=${@:---steps 3 --warmup 2 --settle 0.01 --sectors 360 --no-cpu-baseline --no-end-to-end --no-extras}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p "$OUT"
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d "$OUT/p" -- python3 bench.py $ARGS > "$OUT/p.log" 2>&1 || echo "pass failed"
python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
grep -E "^==|SQ_|TCC_|FETCH|WRITE" "$OUT/summary.txt"
