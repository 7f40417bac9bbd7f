#!/usr/bin/env python3
"""profiles/rNN/busy.json from a tools/pmc_summary.py text summary: what the dominant launch keeps busy besides HBM.

  python tools/make_busy.py gpurun_out/r05/pmc/summary.txt fused_chain_1024x512 > profiles/r05/busy.json

  kernel cycles = GRBM_GUI_ACTIVE / XCDs           (the counter is summed over the 8 XCDs; one launch)
  valu_busy     = SQ_INSTS_VALU x 2 cycles / (SIMDs x kernel cycles)      a wave's vector instruction occupies its SIMD for 2 cycles
                                                                           (MI355X_MICROARCH.md, wave scheduling; 256 CUs x 4 SIMDs)
  lds_busy      = SQ_LDS_IDX_ACTIVE / (CUs x kernel cycles)
bench.py reports them in its `roofline` block only when the fingerprint recorded here is the one of the sources it runs."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
XCDS, CUS, SIMDS = 8, 256, 1024


def main():
    import wrp_amd
    path, kernel = sys.argv[1], sys.argv[2]
    cur, vals = None, {}
    for line in open(path):
        m = re.match(r"== (\S.*)$", line)
        if m:
            cur = m.group(1).strip()
            continue
        m = re.match(r"\s+(\S+)\s+mean/dispatch\s+([0-9.]+)", line)
        if m and cur and kernel in cur and not cur.startswith("kernel durations"):
            vals.setdefault(cur, {})[m.group(1)] = float(m.group(2))
    # the instantiation with the most dispatches is the launch the bench times
    name = max(vals, key=lambda k: len(vals[k]))
    v = vals[name]
    cycles = v["GRBM_GUI_ACTIVE"] / XCDS
    out = {"source": f"{path} (rocprofv3 --pmc, separate passes, tools/profile_pmc.sh; tools/make_busy.py)",
           "fingerprint": wrp_amd.source_fingerprint(), "kernel": name, "kernel_cycles": round(cycles),
           "valu_busy": round(v["SQ_INSTS_VALU"] * 2 / (SIMDS * cycles), 4), "lds_busy": round(v["SQ_LDS_IDX_ACTIVE"] / (CUS * cycles), 4),
           "SQ_INSTS_VALU": v["SQ_INSTS_VALU"], "SQ_LDS_IDX_ACTIVE": v["SQ_LDS_IDX_ACTIVE"], "SQ_INSTS_LDS": v.get("SQ_INSTS_LDS"),
           "SQ_WAIT_ANY_share_of_wave_cycles": round(v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 4) if "SQ_WAIT_ANY" in v and "SQ_WAVE_CYCLES" in v else None}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
