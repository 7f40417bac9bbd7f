#!/usr/bin/env python3
"""Where the fixed cost of a fused launch goes (an experimental stamps build that also records, per workgroup, its entry
into the kernel, the end of the team meeting and its exit: slots [3][8], [4][8], [5][8] of gpurun_out/stamps.npy, written by
tools/fused_stamps.py run with WRP_LIB_PATH on that build).  All times in us from the first workgroup's entry."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
st = np.load(os.path.join(ROOT, "gpurun_out", "stamps.npy"))
S = int(sys.argv[1]) if len(sys.argv) > 1 else 360
t = st.astype(np.float64) / 100.0
ident = st[:, 0, 8]
kind = (ident >> np.uint64(32)).astype(int)
xcc = ((ident >> np.uint64(16)) & np.uint64(0xffff)).astype(int)
entry, joined, leave = t[:, 3, 8], t[:, 4, 8], t[:, 5, 8]
t0 = entry.min()
pr = lambda name, x: print(f"  {name:58s} median {np.median(x):8.2f}   min {x.min():8.2f}   max {x.max():8.2f}")
print(f"{len(st)} workgroups, {S} sectors; whole launch by the stamps (last exit - first entry): {leave.max() - t0:.2f} us")
pr("entry into the kernel", entry - t0)
pr("team meeting over", joined - t0)
tile, row = kind == 0, kind == 1
pr("tile: stage 1 of task 0 starts (input requested, tables)", t[tile, 0, 0] - t0)
pr("tile: A1 of task 0", t[tile, 0, 1] - t0)
pr("tile: A1 of task 1", t[tile, 1, 1] - t0)
pr("row : half 0 of task 0 stored by all tiles", t[row, 0, 1] - t0)
pr("row : rows of half 0 of task 0 transformed", t[row, 0, 3] - t0)
per = np.diff(t[tile][:, 3:16, 0], axis=1)
period = np.median(per)
tasks = 2 * S // 8
print(f"  steady task period {period:.3f} us x {tasks} tasks = {period * tasks:.1f} us")
print("  per team: first A1, exit of its last workgroup, (exit - first stage 1) / tasks")
for x in range(8):
    a = t[tile & (xcc == x), 0, 0].min() - t0
    e = leave[xcc == x].max() - t0
    et = leave[tile & (xcc == x)].max() - t0
    print(f"    xcc {x}: starts {a:7.2f}   tile members out {et:8.2f}   all out {e:8.2f}   per task {(e - a) / tasks:.3f}")
ends = np.array([leave[xcc == x].max() for x in range(8)]) - t0
print(f"  teams finish {ends.min():.2f} .. {ends.max():.2f}: spread {ends.max() - ends.min():.2f} us; mean {ends.mean():.2f}")
