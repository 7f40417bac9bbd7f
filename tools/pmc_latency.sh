#!/bin/bash
# round 5: what a memory request costs inside the fused launches, with and without their input: average L1 -> L2 read / write
# latency (TCP_TCC_*_REQ_LATENCY / TCP_TCC_*_REQ), average L2 -> memory read occupancy (TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ = the mean
# time of a request at the fabric, Little's law), and the L1's stall counters -- for the product library and for tools/make_floor.sh's
# build whose input descriptors have zero records.  usage: tools/pmc_latency.sh OUTDIR   (after tools/make_floor.sh build)
set -u
OUT=${1:-gpurun_out/pmc_latency}
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
ARGS="--steps 3 --warmup 2 --settle 0.01 --sectors 360 --no-cpu-baseline --no-end-to-end --no-extras"
for shape in A B; do
  for lib in product noinput; do
    d=$OUT/${shape}_$lib; mkdir -p $d
    export WRP_LIB_PATH=$ROOT/build/floor/libwrp_$lib.so
    tools/pmc_one.sh $d/l1 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" --shape $shape $ARGS > $d/l1.log 2>&1
    tools/pmc_one.sh $d/ea "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum" --shape $shape $ARGS > $d/ea.log 2>&1
    tools/pmc_one.sh $d/st "TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_ADDR_STALL_CYCLES_sum TA_BUSY_avr GRBM_GUI_ACTIVE" --shape $shape $ARGS > $d/st.log 2>&1
    echo "== shape $shape, $lib"
    cat $d/l1/summary.txt $d/ea/summary.txt $d/st/summary.txt | grep -E "^== fused|TCP_|TCC_|TA_BUSY|GRBM" 
  done
done
