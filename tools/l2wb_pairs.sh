#!/bin/bash
# round 5: does an XCD's L2 keep the rewritten 1 MiB buffer beside a non-temporal stream whose every 128-byte line is asked for
# TWICE (64-byte halves by two workgroups: the 2048 x 128 launch) as well as beside one that asks for whole lines once
# (1024 x 512)?  WRITE_SIZE / FETCH_SIZE of tools/l2wb.hip mode 7 (stores + sc1 read-back + stream, chunks of an XCD contiguous),
# 32 KiB per workgroup = 1 MiB per XCD, 128 KiB streamed per workgroup and repetition = 4 MiB per XCD, 50 repetitions.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/l2wb_pairs; mkdir -p $OUT
run() {  # name counter args...
  local name=$1 ctr=$2; shift 2
  rocprofv3 --pmc $ctr --output-format csv -d $OUT/$name -- build/tools/l2wb "$@" > $OUT/$name.log 2>&1
  python3 - "$OUT/$name" "$name" "$ctr" <<'PY'
import csv, glob, sys
tot = 0.0
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_rewrite" in row["Kernel_Name"] and row["Counter_Name"] == sys.argv[3]:
            tot += float(row["Counter_Value"])
print(f"{sys.argv[2]:34s} {sys.argv[3]} = {tot / 1024:.1f} MiB")
PY
  grep "us per repetition" $OUT/$name.log
}
echo "rewritten per generation: 8 MiB (1 MiB per XCD) x 50; streamed per repetition: 32 MiB (whole lines: 32 MiB of distinct lines; half lines: the same)"
for aux in 1 6; do
  run w_whole_or_half_aux$aux WRITE_SIZE 7 50 32768 131072 $aux 3
  run f_whole_or_half_aux$aux FETCH_SIZE 7 50 32768 131072 $aux 3
done
