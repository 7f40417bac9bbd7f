#!/usr/bin/env python3
"""Phase anatomy of the fused launch (wrp_debug_fused_stamps, a separate diagnostic instantiation).
Every workgroup stamps its first 16 tasks (s_memrealtime, 100 MHz); slot 8 of task 0 holds
kind << 32 | xcc << 16 | rank.
  tile workgroups: 0 task start, 1 stage 1 done (barrier A1), [quarters 0-1 requested, stages 2-3 of group 0,
                   look A: the slot is free of the previous task's half 1 (wave 0), A2, stores: no stamp here -- any
                   stamp between A1 and the stores makes hipcc spill 30-40 registers], 2 quarter 2 requested + group 1
                   written, 6 half 0 drained and counted, 3 A3, 7 quarter 3 requested, stages 2-3 of group 1 done, look B:
                   the rows have half 0 of THIS task, 4 A4 + stores of half 1 issued
  row workgroups : wave 0 (half 0) slots 0-3, wave 4 (half 1) slots 4-7: + 0 task start, + 1 half stored by
                   all tiles, + 2 rows in registers and counted, + 3 rows transformed
Read the SHARES, not the length: stamps forbid overlaps the real launch has."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import wrp_amd
    from oracle import oracle as O
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    zeros = len(sys.argv) > 2 and sys.argv[2] == "zeros"      # an all-zero sweep: same instructions, no switching in the data paths
    dev = torch.device("cuda", 0)
    pool = np.stack([O.synthetic_sector(k) for k in range(2)])
    d_pool = torch.from_numpy(pool.view(np.float32).reshape(2, -1)).to(dev)
    d_iq = d_pool[torch.arange(S, device=dev) % 2].contiguous()
    if zeros:
        d_iq.zero_()
    d_out = torch.empty((S, 512, 2), dtype=torch.float32, device=dev)
    eng = wrp_amd.Engine(device=0, n_slots=1, n_sectors=1, n_elevations=1)
    nwg = torch.cuda.get_device_properties(0).multi_processor_count * 2
    st = np.zeros((nwg, 16, 9), np.uint64)
    lib = eng.lib
    import time
    t_end = time.perf_counter() + 2.0      # the chip at its working point: two seconds of launches on this data first
    while time.perf_counter() < t_end:
        for _ in range(20):
            eng.process_batch_device(d_iq.data_ptr(), S, d_out.data_ptr())
        eng.check()
    for _ in range(2):
        rc = lib.wrp_debug_fused_stamps(eng.handle, C.c_void_p(d_iq.data_ptr()), S, C.c_void_p(d_out.data_ptr()),
                                        st.ctypes.data_as(C.c_void_p), st.size)
        assert rc == 0, (rc, lib.wrp_last_hip_error(eng.handle))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.save(os.path.join(ROOT, "gpurun_out", "stamps.npy"), st)
    clk = st[:, 1, 8].astype(np.float64) / np.maximum(st[:, 2, 8].astype(np.float64), 1.0) * 100.0     # MHz
    print(f"shader clock over the task loop ({'ALL-ZERO input' if zeros else 'synthetic input'}): median {np.median(clk):.0f} MHz "
          f"(p10 {np.percentile(clk, 10):.0f}, p90 {np.percentile(clk, 90):.0f}); launch {np.median(st[:, 2, 8]) / 100.0:.1f} us")
    ident = st[:, 0, 8]
    kind = (ident >> np.uint64(32)).astype(int)
    xcc = ((ident >> np.uint64(16)) & np.uint64(0xffff)).astype(int)
    print("workgroups per (kind, xcc):", {(k, x): int(((kind == k) & (xcc == x)).sum()) for k in (0, 1) for x in range(8)})
    t = st.astype(np.float64) / 100.0     # us
    tasks = min(16, 2 * (S // 8))
    r = slice(3, tasks)
    names = {0: ["stage 1 .. A1", "A1 .. group 1 in LDS (Q0, stage 2, Q1, stage 3, look A, A2, stores, Q2, write)",
                 "drain half 0, count .. A3", "A3 .. A4 + stores (Q3, stage 2, stage 3, look B = wait for the rows of half 0, A4, stores)"],
             1: ["half 0: wait stored", "half 0: row loads + count", "half 0: two row transforms",
                 "(wave 0 -> wave 4)", "half 1: wait stored", "half 1: row loads + count", "half 1: two row transforms"]}
    for k, label in ((0, "tile"), (1, "row")):
        sel = kind == k
        d = np.diff(t[sel][:, r, :len(names[k]) + 1], axis=2)
        if k == 0:
            x = t[sel][:, r, :]
            print(f"    (A1 -> stages 2-3 + look A + A2 + stores + Q2 + write {np.median(x[:, :, 2] - x[:, :, 1]):.2f}, "
                  f"drain + count {np.median(x[:, :, 6] - x[:, :, 2]):.2f}, barrier A3 {np.median(x[:, :, 3] - x[:, :, 6]):.2f}, "
                  f"A3 -> stages 2-3 + look B {np.median(x[:, :, 7] - x[:, :, 3]):.2f}, A4 + stores {np.median(x[:, :, 4] - x[:, :, 7]):.2f})")
        print(f"{label} workgroups ({int(sel.sum())}), tasks 3..{tasks - 1}; median (p10 .. p90) us")
        for i, nm in enumerate(names[k]):
            x = d[:, :, i].ravel()
            print(f"    {nm:42s} {np.median(x):7.2f}  ({np.percentile(x, 10):6.2f} .. {np.percentile(x, 90):6.2f})")
        if k == 1:   # when, relative to the tile workgroups' A1 / A3 of the same task, do the rows get loaded
            tt = t[kind == 0][:, r, :]
            tr = t[sel][:, r, :]
            for x in range(8):
                a1 = np.median(tt[xcc[kind == 0] == x][:, :, 1], axis=0)
                a3 = np.median(tt[xcc[kind == 0] == x][:, :, 3], axis=0)
                h0 = np.median(tr[xcc[sel] == x][:, :, 2], axis=0)
                h1 = np.median(tr[xcc[sel] == x][:, :, 6], axis=0)
                if x == 0:
                    print(f"    xcc 0: half 0 loaded - A3 of its task: {np.round(h0 - a3, 2)}")
                    print(f"    xcc 0: half 1 loaded - A1 of the NEXT task: {np.round(h1[:-1] - a1[1:], 2)}")
        per = np.diff(t[sel][:, r, 0], axis=1).ravel()
        print(f"    task period: median {np.median(per):.2f} us (p10 {np.percentile(per, 10):.2f}, p90 {np.percentile(per, 90):.2f})"
              f"  -> {np.median(per) * 2 / 8:.2f} us/sector with 8 teams")
    eng.close()


if __name__ == "__main__":
    main()
