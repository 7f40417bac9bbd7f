#!/bin/bash
# profiles/rNN/floor[_B].json: the fused launch with every request of its input DROPPED by the descriptor (zero records: the
# loads return at once, the results are wrong on purpose) -- what the launch costs when the input costs nothing.  A timing
# build of the CURRENT sources (a copy of csrc/ with the descriptors' sizes set to 0), timed in one process beside the
# product build with tools/ab.py --no-check.  Part 1 (here, needs hipcc): tools/make_floor.sh build; part 2 (GPU box):
# tools/make_floor.sh run rNN
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
L=$ROOT/build/floor
if [ "${1:-}" = build ]; then
  rm -rf $L; mkdir -p $L/pkg/csrc $L/include
  cp $ROOT/include/wrp.h $L/include/; cp $ROOT/weather-radar-processing_amd/csrc/* $L/pkg/csrc/
  # every descriptor of the INPUT (tile loads of both launches and both formats, the row waves' touches): never valid
  sed -i -E '/make_rsrc\((src|sector_raw|range), valid \?/s/valid \?/false \&\& valid ?/' $L/pkg/csrc/wrp_fused.h $L/pkg/csrc/wrp_fused_b.h
  [ "$(cat $L/pkg/csrc/wrp_fused.h $L/pkg/csrc/wrp_fused_b.h | grep -c "false && valid")" = 6 ] || { echo "make_floor.sh: the no-input edits no longer match the sources"; exit 1; }
  FLAGS=$(grep '^HIPFLAGS' $ROOT/Makefile | sed 's/.*?= //; s/\$(ARCH)/gfx950/')
  /opt/rocm/bin/hipcc $FLAGS -shared -o $L/libwrp_noinput.so $L/pkg/csrc/wrp_engine.hip
  # ... and the 1024 x 512 launch with its input SERVED BY THE L2: the same sixteen requests per lane and task, but every task reads
  # the same 512 KiB (rows p0 + 64 (R & 1) of sector 0's HH plane at the member's column tile of the task: misses the L1, hits the L2)
  rm -rf $L/hit; mkdir -p $L/hit/pkg/csrc $L/hit/include
  cp $ROOT/include/wrp.h $L/hit/include/; cp $ROOT/weather-radar-processing_amd/csrc/* $L/hit/pkg/csrc/
  sed -i -e 's/buf_load_f4<FUSED_INPUT_AUX>(rs, voff, 64 \* R \* DP_N \* 8)/buf_load_f4<FUSED_INPUT_AUX>(rs, voff, 64 * (R \& 1) * DP_N * 8)/' \
         -e 's/buf_load_f4<FUSED_INPUT_AUX>(rs, voff, 64 \* r \* DP_N \* 8)/buf_load_f4<FUSED_INPUT_AUX>(rs, voff, 64 * (r \& 1) * DP_N * 8)/' \
         -e 's/auto tile_src = \[&\](int q) { return iq + ((size_t)(trank + (q >> 1) \* teams) \* channels + (q & 1)) \* RP_M \* (size_t)n; };/auto tile_src = [\&](int q) { return iq + 0 * q; };/' $L/hit/pkg/csrc/wrp_fused.h
  [ "$(grep -c "(R & 1)\|(r & 1) \* DP_N\|iq + 0 \* q" $L/hit/pkg/csrc/wrp_fused.h)" = 3 ] || { echo "make_floor.sh: the L2-hit edits of wrp_fused.h no longer match the sources"; exit 1; }
  # the same for the 2048 x 128 launch: rows p0 + 128 (r & 1) of sector 0 (256 KiB per channel), the touches as well
  sed -i -e 's/buf_load_f4<FUSED_B_INPUT_AUX>(rs, voff, 128 \* R \* row_stride)/buf_load_f4<FUSED_B_INPUT_AUX>(rs, voff, 128 * (R \& 1) * row_stride)/' \
         -e 's/buf_load_f4<FUSED_B_INPUT_AUX>(rs, voff, 128 \* r \* row_stride)/buf_load_f4<FUSED_B_INPUT_AUX>(rs, voff, 128 * (r \& 1) * row_stride)/' \
         -e 's/return iq + ((size_t)(trank + q \* teams) \* channels + ch) \* RB_M \* (size_t)RB_N;/return iq + (size_t)ch * RB_M * (size_t)RB_N;/' \
         -e 's/(size_t)(trank + (q >= 0 \&\& q < tasks ? q : 0) \* teams) \* (RAW/(size_t)0 * (RAW/' $L/hit/pkg/csrc/wrp_fused_b.h
  [ "$(grep -c "(R & 1) \* row_stride\|(r & 1) \* row_stride\|iq + (size_t)ch \* RB_M\|(size_t)0 \* (RAW" $L/hit/pkg/csrc/wrp_fused_b.h)" = 4 ] || { echo "make_floor.sh: the L2-hit edits of wrp_fused_b.h no longer match the sources"; exit 1; }
  /opt/rocm/bin/hipcc $FLAGS -shared -o $L/libwrp_l2hit.so $L/hit/pkg/csrc/wrp_engine.hip
  cp $ROOT/weather-radar-processing_amd/lib/libwrp.so $L/libwrp_product.so
  ls -la $L/*.so
else
  R=${2:-r05}; OUT=$ROOT/gpurun_out/$R; mkdir -p $OUT
  cd $ROOT
  python3 tools/ab.py $L/libwrp_product.so $L/libwrp_noinput.so $L/libwrp_l2hit.so --rounds 20 --no-check > $OUT/floor_A.log 2>&1
  python3 tools/ab.py $L/libwrp_product.so $L/libwrp_noinput.so $L/libwrp_l2hit.so --rounds 20 --no-check --shape B > $OUT/floor_B.log 2>&1
  python3 - $OUT <<'PY'
import json, re, sys, os
sys.path.insert(0, os.getcwd())
import wrp_amd
out = sys.argv[1]
for shape, name in (("A", "floor.json"), ("B", "floor_B.json")):
    t = open(f"{out}/floor_{shape}.log").read()
    med = dict(re.findall(r"(libwrp_\w+)\.so\s+ok=\w+\s+median\s+([0-9.]+)", t))
    rec = {"source": f"tools/make_floor.sh ({out}/floor_{shape}.log): tools/ab.py --no-check, the product build and timing builds of the same sources: input descriptors with zero records; (1024 x 512) every request of the input served by the L2",
           "fingerprint": wrp_amd.source_fingerprint(), "us_per_sector": float(med["libwrp_product"]),
           "no_input_us_per_sector": float(med["libwrp_noinput"])}
    if "libwrp_l2hit" in med:
        rec["l2_hit_input_us_per_sector"] = float(med["libwrp_l2hit"])
    json.dump(rec, open(f"{out}/{name}", "w"), indent=1)
    print(shape, med)
PY
fi
