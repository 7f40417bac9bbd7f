#!/usr/bin/env python3
"""Phase anatomy of the fused 2048 x 128 launch (its STAMPS instantiation, through wrp_debug_fused_stamps).
Wave 0 of every workgroup stamps its first 16 tasks (s_memrealtime, 100 MHz); slot 8 of task 0 = kind << 32 | xcc << 16 | rank.
  tile: 0 task start, 5 column 0 of stage 1 done (the first use of the tile: includes the wait for its last pieces), 1 A1,
        2 stages 2-3 of group 0 + look A + A2 + stores + group 1 written, 6 stores drained (vmcnt), 3 A3,
        7 stages 2-3 of group 1, 4 look B passed; the next 0 = A4 + stores issued
  row : per half g, slots 4 g + : 0 start, 1 half noticed, 2 rows in registers + flag, 3 transformed + products written
usage: fused_stamps_b.py [sectors per launch]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import wrp_amd
    from oracle import oracle as O
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 360
    m, n = 2048, 128
    dev = torch.device("cuda", 0)
    pool = np.stack([O.synthetic_sector(k, m, n) for k in range(4)])
    d_pool = torch.from_numpy(pool.view(np.float32).reshape(4, -1)).to(dev)
    d_iq = d_pool[torch.arange(S, device=dev) % 4].contiguous()
    d_out = torch.empty((S, m // 2, 2), dtype=torch.float32, device=dev)
    eng = wrp_amd.Engine(device=0, n_slots=1, n_sectors=1, n_elevations=1, m=m, n=n)
    nwg = torch.cuda.get_device_properties(0).multi_processor_count * 2
    st = np.zeros((nwg, 16, 9), np.uint64)
    t_end = time.perf_counter() + 2.0
    while time.perf_counter() < t_end:
        for _ in range(20):
            eng.process_batch_device(d_iq.data_ptr(), S, d_out.data_ptr())
        eng.check()
    for _ in range(2):
        rc = eng.lib.wrp_debug_fused_stamps(eng.handle, C.c_void_p(d_iq.data_ptr()), S, C.c_void_p(d_out.data_ptr()),
                                            st.ctypes.data_as(C.c_void_p), st.size)
        assert rc == 0, (rc, eng.lib.wrp_last_hip_error(eng.handle))
    assert st[:, 0, 8].any(), "no stamps came back"
    ident = st[:, 0, 8]
    kind = (ident >> np.uint64(32)).astype(int)
    clk = st[:, 1, 8].astype(np.float64) / np.maximum(st[:, 2, 8].astype(np.float64), 1.0) * 100.0
    print(f"shader clock over the task loop: median {np.median(clk):.0f} MHz; loop {np.median(st[:, 2, 8]) / 100.0:.1f} us for {S} sectors")
    t = st.astype(np.float64) / 100.0
    tasks = min(16, S // 8)
    r = slice(3, tasks - 1)
    x = t[kind == 0]
    nxt = x[:, 4:tasks, 0]
    x = x[:, r, :]
    def med(a):
        a = a.ravel()
        return f"{np.median(a):6.2f}  ({np.percentile(a, 10):5.2f} .. {np.percentile(a, 90):5.2f})"
    print(f"tile workgroups ({int((kind == 0).sum())}), tasks 3..{tasks - 2}; median (p10 .. p90) us")
    print("    start -> column 0 of stage 1 done          ", med(x[:, :, 5] - x[:, :, 0]))
    print("    -> A1 (drain + count of half 1, column 1)   ", med(x[:, :, 1] - x[:, :, 5]))
    print("    -> group 1 written (stages 2-3, look A, A2, stores) ", med(x[:, :, 2] - x[:, :, 1]))
    print("    -> stores of half 0 drained                 ", med(x[:, :, 6] - x[:, :, 2]))
    print("    -> A3                                       ", med(x[:, :, 3] - x[:, :, 6]))
    print("    -> stages 2-3 of group 1 done               ", med(x[:, :, 7] - x[:, :, 3]))
    print("    -> look B passed                            ", med(x[:, :, 4] - x[:, :, 7]))
    print("    -> next task (A4, stores issued)            ", med(nxt[:, :x.shape[1]] - x[:, :, 4]))
    per = np.diff(t[kind == 0][:, 3:tasks, 0], axis=1)
    print(f"    task period {med(per)}  -> {np.median(per) / 8:.3f} us/sector with 8 teams")
    y = t[kind == 1][:, r, :]
    print(f"row workgroups ({int((kind == 1).sum())}), wave 0")
    for g in range(2):
        b = 4 * g
        print(f"    half {g}: wait for the notice {med(y[:, :, b + 1] - y[:, :, b])}   rows + flag {med(y[:, :, b + 2] - y[:, :, b + 1])}"
              f"   transform {med(y[:, :, b + 3] - y[:, :, b + 2])}")
    # one timeline: every stamp relative to the start of the same task in the tile members of the same XCD
    xcc = ((ident >> np.uint64(16)) & np.uint64(0xffff)).astype(int)
    rel_t, rel_r = [], []
    for xc in range(8):
        tt = t[(kind == 0) & (xcc == xc)][:, r, :]
        rr = t[(kind == 1) & (xcc == xc)][:, r, :]
        base = np.median(tt[:, :, 0], axis=0)
        rel_t.append(tt[:, :, :8] - base[None, :, None])
        rel_r.append(rr[:, :, :8] - base[None, :, None])
    rel_t, rel_r = np.concatenate(rel_t), np.concatenate(rel_r)
    def line(a):
        a = a.ravel()
        return f"{np.median(a):6.2f} ({np.percentile(a, 10):6.2f} .. {np.percentile(a, 90):6.2f}, max {a.max():6.2f})"
    print("timeline, us after the median task start of the XCD's tile members: median (p10 .. p90, max over workgroups and tasks)")
    for k, nm in ((0, "tile: task start"), (5, "tile: column 0 of stage 1 done"), (1, "tile: A1"), (2, "tile: group 1 written (stores of half 0 issued)"),
                  (6, "tile: stores of half 0 drained (wave 0)"), (3, "tile: A3 (stored[0] published before it)"), (7, "tile: at look B"), (4, "tile: look B passed")):
        print(f"    {nm:52s} {line(rel_t[:, :, k])}")
    for k, nm in ((0, "row: starts to poll for half 0"), (1, "row: half 0 noticed"), (2, "row: half 0 in registers, loaded[0] published"), (3, "row: half 0 transformed"),
                  (4, "row: starts to poll for half 1 (of this task)"), (5, "row: half 1 noticed"), (6, "row: half 1 in registers, loaded[1] published"), (7, "row: half 1 transformed")):
        print(f"    {nm:52s} {line(rel_r[:, :, k])}")
    np.save(os.path.join(ROOT, "gpurun_out", "stamps_b.npy"), st)
    eng.close()


if __name__ == "__main__":
    main()
