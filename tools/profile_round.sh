#!/bin/bash
# Everything profiles/rNN/ holds, from ONE box and ONE build: bench lines (shape A default command, shape B),
# rocprofv3 --kernel-trace --stats of the same commands, the PMC passes (tools/profile_pmc.sh), the traffic records
# bench.py reads (fingerprint of the sources they were measured on: traffic.json, _B, _W / _W8 = the wire-format launches,
# _BW / _BW8 = the 2048 x 128 launch's), busy.json (valu_busy, lds_busy) and floor.json (the launch without its input:
# tools/make_floor.sh build must have run where hipcc is).  Copy gpurun_out/rNN/ to profiles/rNN/.
# usage: tools/profile_round.sh r05 [pmc|wire|bench]   (one phase per GPU call when the whole does not fit a call's time limit)
set -u
R=${1:-r05}
PHASE=${2:-all}
phase() { [ "$PHASE" = all ] || [ "$PHASE" = "$1" ]; }
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$R; mkdir -p $OUT profiles/$R
T="--steps 3 --warmup 2 --settle 0.01 --sectors 360 --no-cpu-baseline --no-end-to-end --no-extras"
if phase pmc; then
echo "== kernel trace A"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_A -- python3 bench.py --no-cpu-baseline --no-end-to-end --no-extras > $OUT/trace_A.log 2>&1
echo "== kernel trace B"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_B -- python3 bench.py --shape B --no-cpu-baseline --no-end-to-end --no-extras > $OUT/trace_B.log 2>&1
echo "== PMC passes A"; tools/profile_pmc.sh $OUT/pmc > $OUT/pmc.log 2>&1; tail -5 $OUT/pmc.log
python3 tools/make_traffic.py $OUT/pmc/summary.txt 360 fused_chain_1024x512 > $OUT/traffic.json && cp $OUT/traffic.json profiles/$R/traffic.json
python3 tools/make_busy.py $OUT/pmc/summary.txt fused_chain_1024x512 > $OUT/busy.json && cp $OUT/busy.json profiles/$R/busy.json
echo "== PMC passes B"; tools/profile_pmc.sh $OUT/pmc_B --shape B $T > $OUT/pmc_B.log 2>&1; tail -3 $OUT/pmc_B.log
python3 tools/make_traffic.py $OUT/pmc_B/summary.txt 360 fused_chain_2048x128 > $OUT/traffic_B.json && cp $OUT/traffic_B.json profiles/$R/traffic_B.json
python3 tools/make_busy.py $OUT/pmc_B/summary.txt fused_chain_2048x128 > $OUT/busy_B.json && cp $OUT/busy_B.json profiles/$R/busy_B.json
fi
# the wire-format launches: FETCH_SIZE / WRITE_SIZE only; the kernel is named by its template arguments
wire() {  # tag  WRP_BENCH_WIRE  kernel-substring  [bench args]
  local tag=$1 wb=$2 kern=$3; shift 3
  echo "== traffic passes, $tag"; WRP_BENCH_WIRE=$wb tools/profile_traffic.sh $OUT/pmc_$tag "$@" $T > $OUT/pmc_$tag.log 2>&1; tail -3 $OUT/pmc_$tag.log
  python3 tools/make_traffic.py $OUT/pmc_$tag/summary.txt 360 "$kern" > $OUT/traffic_$tag.json && cp $OUT/traffic_$tag.json profiles/$R/traffic_$tag.json
}
if phase wire; then
wire W 12 "fused_chain_1024x512<7, false, 12, false>"
wire W8 8 "fused_chain_1024x512<7, false, 8, false>"
wire BW 12 "fused_chain_2048x128<7, false, false, 12>" --shape B
wire BW8 8 "fused_chain_2048x128<7, false, false, 8>" --shape B
fi
if phase bench; then
echo "== floors (the launches without their input)"; tools/make_floor.sh run $R > $OUT/floor.log 2>&1; tail -2 $OUT/floor.log
cp $OUT/floor.json $OUT/floor_B.json profiles/$R/ 2>/dev/null
echo "== bench B with the records of this build"; python3 bench.py --shape B --no-cpu-baseline > $OUT/bench_B.json 2> $OUT/bench_B.err; tail -c 300 $OUT/bench_B.json
echo "== bench A with the records of this build"; python3 bench.py > $OUT/bench_A.json 2> $OUT/bench_A.err; cat $OUT/bench_A.json
echo "== the driver's command"; python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_command.json 2> $OUT/bench_driver_command.err; tail -c 300 $OUT/bench_driver_command.json
fi
# keep the merged directory small: the stats tables, not the raw traces
find $OUT -name "*_kernel_trace.csv" -size +2M -delete
find $OUT -name "*_agent_info.csv" -delete
