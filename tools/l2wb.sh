#!/bin/bash
# rocprofv3 PMC passes around tools/l2wb (build/tools/l2wb): WRITE_SIZE / FETCH_SIZE per configuration
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/l2wb; mkdir -p $OUT
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
print(f"{sys.argv[2]:28s} {sys.argv[3]} = {tot / 1024:.1f} MiB")
PY
}
for st in 0 3; do
  run f_m4_c32768_st$st FETCH_SIZE 4 50 32768 131072 1 $st
  run w_m4_c32768_st$st WRITE_SIZE 4 50 32768 131072 1 $st
  run f_m7_c65536_st$st FETCH_SIZE 7 50 65536 131072 1 $st
  run w_m7_c65536_st$st WRITE_SIZE 7 50 65536 131072 1 $st
done
