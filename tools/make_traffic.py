#!/usr/bin/env python3
"""profiles/rNN/traffic.json from a tools/pmc_summary.py text summary (FETCH_SIZE and WRITE_SIZE
passes, one dispatch = one chunk of `--sectors` sectors):

  python tools/make_traffic.py profiles/r01/pmc_summary_360sectors_default.txt 360 > profiles/r01/traffic.json

HBM bytes per launch = 2 x FETCH_SIZE (KiB; gfx950 tallies a 128-byte read request as 64 bytes,
MI355X_MICROARCH.md, HBM section) + WRITE_SIZE (KiB), summed over the kernels of the chain."""
import json
import re
import sys


def main():
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
           "sectors_per_launch": sectors,
           "correction": "FETCH_SIZE doubled (gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md HBM section), WRITE_SIZE as read"}
    total = 0.0
    for k, v in kernels.items():
        if "FETCH_SIZE_KiB" not in v or "WRITE_SIZE_KiB" not in v:
            continue
        v["bytes_corrected"] = (2 * v["FETCH_SIZE_KiB"] + v["WRITE_SIZE_KiB"]) * 1024
        out[k] = v
    chain = [k for k in out if k.startswith(("range_pass", "doppler_pass"))]
    total = sum(out[k]["bytes_corrected"] for k in chain)
    out["chain"] = chain
    out["bytes_per_launch_pair"] = total
    out["bytes_per_sector"] = total / sectors
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
