#!/usr/bin/env python3
"""Phase anatomy of the fused dataflow launch (wrp_debug_fused_stamps).  Every workgroup stamps
its first 16 items: slot 0 start, 1 (A: stages 1-2 done | B: tiles of the task complete),
2 (A: mid buffer free | B: rows transformed), 4 end, 5 = 1000*isB + task, 6 shader clock."""
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
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    dev = torch.device("cuda", 0)
    pool = np.stack([O.synthetic_sector(k) for k in range(2)])
    d_pool = torch.from_numpy(pool.view(np.float32).reshape(2, -1)).to(dev)
    d_iq = d_pool[torch.arange(S, device=dev) % 2].contiguous()
    d_out = torch.empty((S, 512, 2), dtype=torch.float32, device=dev)
    eng = wrp_amd.Engine(device=0, n_slots=1, n_sectors=1, n_elevations=1, flags=0x100)
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    st = np.zeros((ncu, 16, 8), np.uint64)
    lib = eng.lib
    lib.wrp_debug_fused_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
    for _ in range(2):
        rc = lib.wrp_debug_fused_stamps(eng.handle, C.c_void_p(d_iq.data_ptr()), S, C.c_void_p(d_out.data_ptr()),
                                        st.ctypes.data_as(C.c_void_p), st.size)
        assert rc == 0, rc
    used = st[:, :, 4] != 0
    kind = st[:, :, 5] >= 1000
    t = st.astype(np.float64) / 100.0     # us
    clk = np.diff(st[:, :, 6].astype(np.float64), axis=1) / np.maximum(np.diff(t[:, :, 0], axis=1), 1e-9)
    ok = used[:, 1:] & used[:, :-1]
    print(f"shader clock during the launch: median {np.median(clk[ok]):.0f} MHz; "
          f"{int(used.sum())} items recorded on {ncu} workgroups ({int((used & kind).sum())} B)")
    skip = np.arange(16)[None, :] >= 2        # steady state only
    for nm, sel, names, d in (
            ("A item (range tile)", used & ~kind & skip,
             ("wait tile arrival", "stages 1-2 (+pop, prefetch issue)", "wait mid free", "stage 3", "drain + count"),
             [t[:, :, 7] - t[:, :, 0], t[:, :, 1] - t[:, :, 7], t[:, :, 2] - t[:, :, 1], t[:, :, 3] - t[:, :, 2], t[:, :, 4] - t[:, :, 3]]),
            ("B item (16 rows)", used & kind & skip,
             ("wait tiles complete", "load + transform rows (wave 0)", "the other 15 waves", "wait order", "publish + count"),
             [t[:, :, 1] - t[:, :, 0], t[:, :, 2] - t[:, :, 1], t[:, :, 7] - t[:, :, 2], t[:, :, 3] - t[:, :, 7], t[:, :, 4] - t[:, :, 3]])):
        if not sel.any():
            continue
        tot = (t[:, :, 4] - t[:, :, 0])[sel]
        print(f"{nm}: median {np.median(tot):.2f} us (p10 {np.percentile(tot, 10):.2f}, p90 {np.percentile(tot, 90):.2f}), n={sel.sum()}")
        for x, n2 in zip(d, names):
            x = x[sel]
            print(f"    {n2:34s} {np.median(x):7.2f}  ({np.percentile(x, 10):6.2f} .. {np.percentile(x, 90):6.2f})")
    # who is slow?  per workgroup: time in its own arithmetic vs time waiting for others (items 2..)
    own = np.where(kind, t[:, :, 2] - t[:, :, 1], (t[:, :, 1] - t[:, :, 7]) + (t[:, :, 3] - t[:, :, 2]))
    wait = np.where(kind, (t[:, :, 1] - t[:, :, 0]) + (t[:, :, 3] - t[:, :, 2]), (t[:, :, 7] - t[:, :, 0]) + (t[:, :, 2] - t[:, :, 1]))
    own_wg = np.where(used & skip, own, 0).sum(axis=1)
    wait_wg = np.where(used & skip, wait, 0).sum(axis=1)
    order = np.argsort(wait_wg)
    print(f"per workgroup over items 2..15: own arithmetic median {np.median(own_wg):.1f} us (min {own_wg.min():.1f}, max {own_wg.max():.1f}); "
          f"waiting median {np.median(wait_wg):.1f} us (min {wait_wg.min():.1f}, max {wait_wg.max():.1f})")
    print("  least-waiting workgroups (the pace setters): " +
          ", ".join(f"wg{w}: own {own_wg[w]:.1f} wait {wait_wg[w]:.1f}" for w in order[:6]))
    np.save(os.path.join(ROOT, "gpurun_out", "stamps.npy"), st)
    gap = (t[:, 1:, 0] - t[:, :-1, 4])[ok]
    print(f"gap between items: median {np.median(gap):.2f} us")
    last = np.where(used, t[:, :, 4], 0).max(axis=1)
    first = np.where(used, t[:, :, 0], np.inf).min(axis=1)
    n_items = used.sum(axis=1)
    print(f"items per workgroup in the stamped window: {n_items.min()}..{n_items.max()}, "
          f"us per item per workgroup: median {np.median((last - first) / np.maximum(n_items, 1)):.2f}")
    eng.close()


if __name__ == "__main__":
    main()
