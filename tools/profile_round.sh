#!/bin/bash
# Everything profiles/rNN/ holds, from ONE box and ONE build: bench lines (shape A default command, shape B),
# rocprofv3 --kernel-trace --stats of the same commands, the PMC passes (tools/profile_pmc.sh) and the
# traffic.json bench.py reads (fingerprint of the sources it was measured on).  Copy gpurun_out/rNN/ to profiles/rNN/.
# usage: tools/profile_round.sh r04
set -u
R=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$R; mkdir -p $OUT
echo "== bench A (default command)"; python3 bench.py > $OUT/bench_A_first.json 2> $OUT/bench_A_first.err; tail -c 400 $OUT/bench_A_first.json
echo "== kernel trace A"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_A -- python3 bench.py --no-cpu-baseline --no-end-to-end --no-extras > $OUT/trace_A.log 2>&1
echo "== bench B"; python3 bench.py --shape B --no-cpu-baseline > $OUT/bench_B.json 2> $OUT/bench_B.err; tail -c 300 $OUT/bench_B.json
echo "== kernel trace B"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_B -- python3 bench.py --shape B --no-cpu-baseline --no-end-to-end --no-extras > $OUT/trace_B.log 2>&1
echo "== PMC passes"; tools/profile_pmc.sh $OUT/pmc > $OUT/pmc.log 2>&1; tail -5 $OUT/pmc.log
python3 tools/make_traffic.py $OUT/pmc/summary.txt 360 fused_chain_1024x512 > $OUT/traffic.json && mkdir -p profiles/$R && cp $OUT/traffic.json profiles/$R/traffic.json
echo "== traffic passes, shape B"; tools/profile_traffic.sh $OUT/pmc_B --shape B --steps 3 --warmup 2 --settle 0.01 --sectors 360 --no-cpu-baseline --no-end-to-end --no-extras > $OUT/pmc_B.log 2>&1; tail -3 $OUT/pmc_B.log
python3 tools/make_traffic.py $OUT/pmc_B/summary.txt 360 fused_chain_2048x128 > $OUT/traffic_B.json && cp $OUT/traffic_B.json profiles/$R/traffic_B.json
echo "== traffic passes, wire-format input"; WRP_BENCH_WIRE=1 tools/profile_traffic.sh $OUT/pmc_W --steps 3 --warmup 2 --settle 0.01 --sectors 360 --no-cpu-baseline --no-end-to-end --no-extras > $OUT/pmc_W.log 2>&1; tail -3 $OUT/pmc_W.log
python3 tools/make_traffic.py $OUT/pmc_W/summary.txt 360 "false, true" > $OUT/traffic_W.json && cp $OUT/traffic_W.json profiles/$R/traffic_W.json
echo "== bench B with the traffic of this build"; python3 bench.py --shape B --no-cpu-baseline > $OUT/bench_B.json 2> $OUT/bench_B.err; tail -c 300 $OUT/bench_B.json
echo "== bench A with the traffic of this build"; python3 bench.py > $OUT/bench_A.json 2> $OUT/bench_A.err; cat $OUT/bench_A.json
# keep the merged directory small: the stats tables, not the raw traces
find $OUT -name "*_kernel_trace.csv" -size +2M -delete
find $OUT -name "*_agent_info.csv" -delete
