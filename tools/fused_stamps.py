#!/usr/bin/env python3
"""Phase anatomy of the fused launch (wrp_debug_fused_stamps).  Every workgroup stamps its first 16
rounds (s_memrealtime, 100 MHz): 0 round start, 1 tile arrived, 2 stages 1-2 done, 3 stage 3 done and
all tiles of the previous task stored, 4 row in registers (wave 0), 5 row transformed (wave 0),
6 all waves done and all rows of the previous task loaded, 7 tile stores issued."""
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
    dev = torch.device("cuda", 0)
    pool = np.stack([O.synthetic_sector(k) for k in range(2)])
    d_pool = torch.from_numpy(pool.view(np.float32).reshape(2, -1)).to(dev)
    d_iq = d_pool[torch.arange(S, device=dev) % 2].contiguous()
    d_out = torch.empty((S, 512, 2), dtype=torch.float32, device=dev)
    tcols = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    eng = wrp_amd.Engine(device=0, n_slots=1, n_sectors=1, n_elevations=1, flags=0x100 | tcols)
    ncu = torch.cuda.get_device_properties(0).multi_processor_count * (2 if tcols == 8 else 1)
    st = np.zeros((ncu, 16, 8), np.uint64)
    lib = eng.lib
    lib.wrp_debug_fused_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
    for _ in range(2):
        rc = lib.wrp_debug_fused_stamps(eng.handle, C.c_void_p(d_iq.data_ptr()), S, C.c_void_p(d_out.data_ptr()),
                                        st.ctypes.data_as(C.c_void_p), st.size)
        assert rc == 0, rc
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.save(os.path.join(ROOT, "gpurun_out", "stamps.npy"), st)
    t = st.astype(np.float64) / 100.0     # us
    rounds = min(16, 2 * (S // 8))
    r = slice(2, rounds)                   # steady state: both halves of the round present
    names = ["wait tile arrival", "stages 1-2 (+ count, tile request)", "stage 3 + wait tiles stored", "row load (wave 0)",
             "row transform (wave 0)", "other waves + wait rows loaded", "tile stores issued"]
    d = np.diff(t[:, r, :], axis=2)
    print(f"{ncu} workgroups, rounds 2..{rounds - 1}; median (p10 .. p90) us")
    for k, nm in enumerate(names):
        x = d[:, :, k].ravel()
        print(f"    {nm:36s} {np.median(x):7.2f}  ({np.percentile(x, 10):6.2f} .. {np.percentile(x, 90):6.2f})")
    per_round = np.diff(t[:, r, 0], axis=1).ravel()
    print(f"round: median {np.median(per_round):.2f} us (p10 {np.percentile(per_round, 10):.2f}, p90 {np.percentile(per_round, 90):.2f})"
          f"  -> {np.median(per_round) * 2 / 8:.2f} us/sector with 8 teams")
    eng.close()


if __name__ == "__main__":
    main()
