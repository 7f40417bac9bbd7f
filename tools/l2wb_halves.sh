#!/bin/bash
# round 5: the 2048 x 128 launch writes back 19 % of its intermediate (1024 x 512: 1.5 %).  Its tile members store 64-byte HALF lines
# of the slot (8 columns), the other half coming from another CU at another time.  Does that by itself make an XCD's L2 write
# lines back?  tools/l2wb.hip mode 7 (stores + sc1 read-back + non-temporal stream, chunks of an XCD contiguous), 1 MiB per XCD
# rewritten 50 times beside 4 MiB per XCD and repetition of stream: whole lines by one workgroup (st 0), half lines by two
# workgroups (st 4), half lines by one workgroup at two times (st 5); stream of whole lines (aux 1) or paired half lines (aux 6).
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/l2wb_halves; mkdir -p $OUT
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
echo "rewritten per generation: 8 MiB (1 MiB per XCD) x 50 = 400 MiB of stores; streamed per repetition: 32 MiB"
for st in 0 4 5; do for aux in 1 6; do
  run w_st${st}_aux$aux WRITE_SIZE 7 50 32768 131072 $aux $st
  run f_st${st}_aux$aux FETCH_SIZE 7 50 32768 131072 $aux $st
done; done
echo "without the stream (mode 5: stores + read-back):"
for st in 0 4; do run w_st${st}_nostream WRITE_SIZE 5 50 32768 131072 1 $st; done
