#!/bin/bash
# Collect rocprofv3 PMC counters for the bench kernels, one --pmc pass per invocation
# (MI355X_MICROARCH.md "rocprofv3 PMC slots": 8 SQ / 4 TCC slots; FETCH_SIZE costs 3,
# WRITE_SIZE costs 2 -> separate passes).  Never combined with sys/hip tracing.
# usage: tools/profile_pmc.sh <outdir> [bench args...]
set -u
OUT=${1:-gpurun_out/pmc}; shift || true
ARGS=${@:---steps 3 --warmup 2 --settle 0.01 --sectors 360 --no-cpu-baseline --no-end-to-end --no-extras}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p "$OUT"
pass() {
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -- python3 bench.py $ARGS \
      > "$OUT/$name.log" 2>&1 || echo "pass $name failed (see $OUT/$name.log)"
}
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
pass sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
pass sq3 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_THREAD_CYCLES_VALU
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
pass tcc2 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum TCC_TAG_STALL_sum
pass grbm GRBM_GUI_ACTIVE
python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
