#!/bin/bash
# l2wb with the streamed input in ordinary / uncached / fine-grained memory: does the L2 keep the rewritten buffer then?
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
for mem in 0 1 2; do
  for aux in 0 1; do
    build/tools/l2wb 7 50 65536 131072 $aux 3 $mem | head -1
    run f_m7_aux${aux}_mem$mem FETCH_SIZE 7 50 65536 131072 $aux 3 $mem
    run w_m7_aux${aux}_mem$mem WRITE_SIZE 7 50 65536 131072 $aux 3 $mem
  done
done
