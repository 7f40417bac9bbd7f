#!/usr/bin/env python3
"""Phase anatomy of the tile/row fused launch (wrp_fused_roles.h) from wrp_debug_fused_stamps.
Tile workgroups (first half of the grid) stamp their first 16 items (= tiles): 0 start, 1 stages 1-3
computed, 2 buffer free (previous task's rows loaded), 3 stores issued (+ drained and counted for the
second tile of a task).  Row workgroups stamp their first 16 tasks: 0 start, 1 tiles stored,
2 rows in registers, 3 rows transformed (wave 0)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def show(name, x):
    x = np.asarray(x).ravel()
    print(f"    {name:44s} {np.median(x):7.2f}  ({np.percentile(x, 10):6.2f} .. {np.percentile(x, 90):6.2f})")


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
    eng = wrp_amd.Engine(device=0, n_slots=1, n_sectors=1, n_elevations=1, flags=0x100 | 8)
    nwg = torch.cuda.get_device_properties(0).multi_processor_count * 2
    st = np.zeros((nwg, 16, 8), np.uint64)
    lib = eng.lib
    lib.wrp_debug_fused_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
    for _ in range(2):
        rc = lib.wrp_debug_fused_stamps(eng.handle, C.c_void_p(d_iq.data_ptr()), S, C.c_void_p(d_out.data_ptr()),
                                        st.ctypes.data_as(C.c_void_p), st.size)
        assert rc == 0, rc
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.save(os.path.join(ROOT, "gpurun_out", "roles_stamps.npy"), st)
    t = st.astype(np.float64) / 100.0
    kind = st[:, 0, 5]                       # wrp_fused64.h records the kind it took at run time (1 tile, 2 rows)
    if kind.max() > 0:
        tiles, rows = t[kind == 1], t[kind == 2]
        print(f"{(kind == 1).sum()} tile workgroups, {(kind == 2).sum()} row workgroups (kind chosen per physical CU)")
    else:
        tiles, rows = t[: nwg // 2], t[nwg // 2:]
    print("tile workgroups, items 4..15 (us: median, p10 .. p90)")
    show("stages 1-3 of one 8-column tile", tiles[:, 4:, 1] - tiles[:, 4:, 0])
    if st[:, 4:, 7].max() > 0:             # wrp_fused64.h: finer stamps
        show("   wait tile arrival", tiles[:, 4:, 4] - tiles[:, 4:, 0])
        show("   stage 1 + barrier", tiles[:, 4:, 6] - tiles[:, 4:, 4])
        show("   tile request + stage 2 + barrier", tiles[:, 4:, 7] - tiles[:, 4:, 6])
        show("   stage 3 arithmetic", tiles[:, 4:, 1] - tiles[:, 4:, 7])
    show("wait buffer free (first tile of a task)", (tiles[:, 4:, 2] - tiles[:, 4:, 1])[:, 0::2])
    show("barrier only (second tile)", (tiles[:, 4:, 2] - tiles[:, 4:, 1])[:, 1::2])
    show("stores (first tile)", (tiles[:, 4:, 3] - tiles[:, 4:, 2])[:, 0::2])
    show("stores + drain + count (second tile)", (tiles[:, 4:, 3] - tiles[:, 4:, 2])[:, 1::2])
    show("item to item", np.diff(tiles[:, 4:, 0], axis=1))
    print("row workgroups, tasks 2..15")
    show("wait tiles stored", rows[:, 2:, 1] - rows[:, 2:, 0])
    show("load 2 rows per wave", rows[:, 2:, 2] - rows[:, 2:, 1])
    show("transform 2 rows (wave 0)", rows[:, 2:, 3] - rows[:, 2:, 2])
    per_task = np.diff(rows[:, 2:, 0], axis=1)
    show("task to task", per_task)
    print(f"  -> {np.median(per_task) * 2 / 8:.2f} us/sector with 8 teams")
    eng.close()


if __name__ == "__main__":
    main()
