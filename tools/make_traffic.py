#!/usr/bin/env python3
"""profiles/rNN/traffic.json from a tools/pmc_summary.py text summary (FETCH_SIZE and WRITE_SIZE
passes, one dispatch = one launch over `sectors` sectors):

  python tools/make_traffic.py gpurun_out/pmc/summary.txt 360 > profiles/r02/traffic.json

HBM bytes per launch = 2 x FETCH_SIZE (KiB; gfx950 tallies a 128-byte read request as 64 bytes,
MI355X_MICROARCH.md, HBM section) + WRITE_SIZE (KiB), summed over the kernels of the chain.  The
file records the fingerprint of the library sources it was measured on; bench.py reports the
traffic only when that fingerprint is the one of the sources it runs."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import wrp_amd
    path, sectors = sys.argv[1], int(sys.argv[2])
    kernels, cur = {}, None
    for line in open(path):
        m = re.match(r"== (\S.*)$", line)
        if m:
            cur = m.group(1).strip()
            continue
        m = re.match(r"\s+(FETCH_SIZE|WRITE_SIZE)\s+mean/dispatch\s+([0-9.]+)", line)
        if m and cur and not cur.startswith("kernel durations"):
            kernels.setdefault(cur, {})[m.group(1) + "_KiB"] = float(m.group(2))
    out = {"source": f"{path} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, tools/profile_pmc.sh)",
           "fingerprint": wrp_amd.source_fingerprint(),
           "sectors_per_launch": sectors,
           "correction": "FETCH_SIZE doubled (gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md HBM section), WRITE_SIZE as read"}
    for k, v in kernels.items():
        if "FETCH_SIZE_KiB" not in v or "WRITE_SIZE_KiB" not in v:
            continue
        v["bytes_corrected"] = (2 * v["FETCH_SIZE_KiB"] + v["WRITE_SIZE_KiB"]) * 1024
        out[k] = v
    chain = [k for k in out if k.startswith(("fused_chain", "range_pass", "doppler_pass"))]
    if len(sys.argv) > 3:      # the kernel(s) of the chain named explicitly (a run that launched other forms as well)
        chain = [k for k in chain if sys.argv[3] in k]
    out["chain"] = chain
    out["fused"] = any(k.startswith("fused_chain") for k in chain)
    out["bytes_per_launch"] = sum(out[k]["bytes_corrected"] for k in chain)
    out["bytes_per_sector"] = out["bytes_per_launch"] / sectors
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
