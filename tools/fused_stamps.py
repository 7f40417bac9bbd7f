#!/usr/bin/env python3
"""Phase anatomy of the fused persistent launch (wrp_debug_fused_stamps): per channel-task and
workgroup, microseconds spent in  A12 (stages 1-2) | wait mid free | A3 store+drain |
team barrier | B (Doppler rows) | gap to next task (prefetch wait)."""
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
    ntask = min(16, 2 * (S // 8))
    dclk = np.diff(st[:, :ntask, 6].astype(np.float64), axis=1)
    dreal = np.diff(st[:, :ntask, 0].astype(np.float64), axis=1) / 100.0
    print(f"shader clock during the launch: median {np.median(dclk / dreal):.0f} MHz")
    t = st.astype(np.float64) / 100.0     # us
    t = t[:, :ntask, :6]
    ph = np.diff(t, axis=2)                # A12, wait2, A3, bar1, B
    gap = t[:, 1:, 0] - t[:, :-1, 5]       # end of B -> next tile ready
    names = ["A12 stages1-2", "wait mid free", "A3 store+drain", "team barrier", "B doppler rows"]
    print(f"{ntask} tasks/team recorded, {ncu} workgroups; median (p10..p90) us, tasks 2.. only")
    for k, nm in enumerate(names):
        x = ph[:, 2:, k].ravel()
        print(f"  {nm:16s} {np.median(x):7.2f}  ({np.percentile(x, 10):6.2f} .. {np.percentile(x, 90):6.2f})")
    x = gap[:, 2:].ravel()
    print(f"  {'gap/prefetch':16s} {np.median(x):7.2f}  ({np.percentile(x, 10):6.2f} .. {np.percentile(x, 90):6.2f})")
    per_task = (t[:, -1, 5] - t[:, 2, 0]) / (ntask - 3 + 1e-9)
    print(f"  per channel-task {np.median(per_task):.2f} us  -> {np.median(per_task) * 2 / 8:.2f} us/sector with 8 teams")
    eng.close()


if __name__ == "__main__":
    main()
