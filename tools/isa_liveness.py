#!/usr/bin/env python3
"""Approximate VGPR liveness over the main loop of a kernel in hipcc's -S output (straight-line backward pass over the loop
body, three times round for the loop-carried values): the pressure at every barrier, memory instruction and peak.
  python tools/isa_liveness.py /tmp/e.s fused_chain_1024x512ILi7ELb0ELb0ELb0E [.LBB26_115]
How round 4 found that hipcc sinks the last butterfly layer of stage 1's parked values behind barrier A1."""
import re, sys
t = open(sys.argv[1]).read()
key = sys.argv[2]
loop_label = sys.argv[3] if len(sys.argv) > 3 else None
m = re.search(r"^(_ZN3wrp\w*%s\w*):" % key, t, re.M)
i = m.end(); j = t.index(".Lfunc_end", i)
lines = t[i:j].split("\n")
# find the biggest loop: a label with 'Loop Header' and the last backward branch to it
labels = {}
for k, ln in enumerate(lines):
    mm = re.match(r"^(\.LBB\d+_\d+):", ln)
    if mm: labels[mm.group(1)] = k
best = None
for k, ln in enumerate(lines):
    mm = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)|s_branch (\.LBB\d+_\d+)", ln)
    if mm:
        lab = mm.group(1) or mm.group(2)
        if lab in labels and labels[lab] < k:
            if loop_label and lab != loop_label: continue
            if best is None or k - labels[lab] > best[1] - best[0]: best = (labels[lab], k, lab)
s, e, lab = best
print("loop", lab, "lines", s, e)
def regs(op):
    out = []
    for mm in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", op):
        if mm.group(1): out += list(range(int(mm.group(1)), int(mm.group(2)) + 1))
        else: out.append(int(mm.group(3)))
    return out
body = []
for k in range(s, e + 1):
    ln = lines[k].split(";")[0].strip()
    if not ln or ln.startswith(".") or ln.endswith(":"): continue
    parts = ln.split(None, 1)
    op = parts[0]; args = parts[1] if len(parts) > 1 else ""
    ops = [a.strip() for a in args.split(",")]
    nodst = op.startswith(("ds_write", "buffer_store", "scratch_store", "global_store", "s_", "v_cmp", "v_readfirstlane", "v_readlane", "ds_add_u32", "v_cmpx", "buffer_wbl2", "buffer_inv")) and not op.startswith("ds_add_rtn")
    if op.startswith("v_cmp") or op.startswith("v_readfirstlane") or op.startswith("v_readlane"):
        d, u = [], [r for a in ops[1:] for r in regs(a)]
    elif nodst:
        d, u = [], [r for a in ops for r in regs(a)]
    else:
        d = regs(ops[0]) if ops else []
        u = [r for a in ops[1:] for r in regs(a)]
        if op.startswith(("v_fmac", "v_mac", "v_pk_fma")) : u += d
    body.append((k, op, d, u, ln))
# two passes around the loop for loop-carried liveness
live = set()
prof = [0] * len(body)
for _ in range(3):
    for idx in range(len(body) - 1, -1, -1):
        k, op, d, u, ln = body[idx]
        live -= set(d)
        live |= set(u)
        prof[idx] = len(live)
peak = max(prof)
print("peak live VGPRs", peak)
# print profile at markers
for idx, (k, op, d, u, ln) in enumerate(body):
    if op in ("s_barrier",) or "scratch" in op or prof[idx] >= peak - 1 and (idx == 0 or prof[idx - 1] < peak - 1) or op.startswith("buffer_load_dwordx4") or op.startswith("buffer_store"):
        print("%5d %4d  %s" % (k, prof[idx], ln[:90]))
