#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output (one directory per pass) per kernel:
mean counter value per dispatch, for the kernels whose name contains 'wrp::'."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    for k in ("range_pass", "doppler_pass", "decode", "fused"):
        if k in name:
            i = name.index(k)
            return name[i:name.index("(", i)] if "(" in name[i:] else name[i:]
    return None


def main(root):
    acc = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> [values per dispatch]
    dur = defaultdict(list)
    for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = defaultdict(float)
        names = {}
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = short(row.get("Kernel_Name", ""))
                if not k:
                    continue
                key = (row["Dispatch_Id"], row["Counter_Name"])
                per_dispatch[key] += float(row["Counter_Value"])
                names[row["Dispatch_Id"]] = k
        for (d, c), v in per_dispatch.items():
            acc[names[d]][c].append(v)
    for f in glob.glob(os.path.join(root, "*", "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = short(row.get("Kernel_Name", ""))
                if k:
                    dur[(os.path.relpath(f, root).split(os.sep)[0], k)].append(
                        (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    for k in sorted(acc):
        print(f"== {k}")
        for c in sorted(acc[k]):
            v = acc[k][c]
            print(f"   {c:32s} mean/dispatch {sum(v) / len(v):16.1f}   (n={len(v)})")
    print("== kernel durations under profiling (us, mean per dispatch)")
    for (p, k), v in sorted(dur.items()):
        print(f"   {p:8s} {k:40s} {sum(v) / len(v):10.2f}  (n={len(v)})")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc")
