#!/usr/bin/env python3
"""A/B timing of two BUILDS of libwrp.so in ONE process, rounds interleaved (the boxes differ by +-3 % and a run of
tune.py per build by +-1.5 %: too coarse for a 1 % change).

  python tools/ab.py build/exp/libs/libwrp_A.so build/exp/libs/libwrp_B.so [--sectors 360] [--rounds 40] [--iters 10]

Each library is dlopen'ed under its own path (two independent instances), gets one engine on device 0 and the same
input; reports the median us/sector of each and the paired difference B - A with its standard error.
Caveat, measured: two engines of the SAME build differ by by 0.3 % (0.8 % with the counter protocol of mid round 2; standard error 0.05 %) -- where an engine's
slots and control block land in memory matters that much; give the same library twice to see the floor on a box."""
import argparse
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--sectors", type=int, default=360)
    ap.add_argument("--rounds", type=int, default=40)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--shape", choices=["A", "B"], default="A")
    ap.add_argument("--no-check", action="store_true", help="timing-only builds whose results are wrong on purpose")
    ap.add_argument("--wire", action="store_true", help="time wrp_process_batch_raw_device (wire-format input) instead")
    ap.add_argument("--wire8", action="store_true", help="... with 8-byte samples (WRP_FLAG_WIRE_8)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import wrp_amd
    from wrp_amd import binding as B
    from oracle import oracle as O

    dev = torch.device("cuda", 0)
    torch.cuda.init()
    S = args.sectors
    m, n = (1024, 512) if args.shape == "A" else (2048, 128)
    pool = np.stack([O.synthetic_sector(k, m, n) for k in range(4)])
    d_pool = torch.from_numpy(pool.view(np.float32).reshape(4, -1)).to(dev)
    d_iq = d_pool[torch.arange(S, device=dev) % 4].contiguous()
    d_out = torch.empty((S, m // 2, 2), dtype=torch.float32, device=dev)
    want = O.sector(pool[1][0], pool[1][1], dtype=np.float64)

    engines = []
    for path in args.libs:
        B._LIB = None
        B.lib_path = (lambda p: (lambda: p))(os.path.abspath(path))     # the binding's loader, pointed at this build
        e = wrp_amd.Engine(device=0, n_slots=1, n_sectors=1, n_elevations=1, m=m, n=n, flags=wrp_amd.FLAG_WIRE_8 if args.wire8 else 0)
        e.process_batch_device(d_iq.data_ptr(), S, d_out.data_ptr())
        torch.cuda.synchronize()
        got = d_out.cpu().numpy()
        ok = bool(np.max(np.abs(got[1][1:] - want[1:])) < 1e-3)
        if not engines:
            first = got.copy()
        elif not args.no_check:     # the builds are meant to be bit-identical: say so, or say where they are not
            same = np.array_equal(first.view(np.uint32), got.view(np.uint32))
            print(f"{os.path.basename(path)}: output {'bit-identical to' if same else 'DIFFERS from'} {engines[0][0]}"
                  + ("" if same else f" (max |diff| {np.nanmax(np.abs(first - got)):.3g})"))
        engines.append((os.path.basename(path), e, ok, []))
    k = 1e3 / (args.iters * S)
    if args.wire or args.wire8:
        import time
        w = np.zeros((4, m * n, 4 if args.wire8 else 6), dtype=">i2")
        for q in range(4):
            for c in range(2):
                w[q, :, 2 * c] = pool[q][c].real.ravel()
                w[q, :, 2 * c + 1] = pool[q][c].imag.ravel()
        d_w = torch.from_numpy(np.frombuffer(w.tobytes(), np.uint8).reshape(4, -1)).to(dev)
        d_raw = d_w[torch.arange(S, device=dev) % 4].contiguous()

        class Timed:
            def __init__(self, e):
                self.e = e

            def time_batch_device(self, a, S_, o, iters, per_kernel=False):
                self.e.check()
                t0 = time.perf_counter()
                for _ in range(iters):
                    self.e.process_batch_raw_device(d_raw.data_ptr(), S_, o)
                self.e.check()
                return ((time.perf_counter() - t0) * 1e3, None, None)
        engines = [(nm, Timed(e), ok, tt) for nm, e, ok, tt in engines]
    for _ in range(5):                                    # settle
        for _, e, _, _ in engines:
            e.time_batch_device(d_iq.data_ptr(), S, d_out.data_ptr(), args.iters, per_kernel=False)
    for r in range(args.rounds):
        order = engines if r % 2 == 0 else engines[::-1]
        for _, e, _, tt in order:
            t = e.time_batch_device(d_iq.data_ptr(), S, d_out.data_ptr(), args.iters, per_kernel=False)
            tt.append((t[0] if isinstance(t, tuple) else t) * k)
    base = engines[0][3]
    for name, e, ok, tt in engines:
        d = [b - a for a, b in zip(base, tt)]
        sem = statistics.pstdev(d) / len(d) ** 0.5
        print(f"{name:28s} ok={ok}  median {statistics.median(tt):6.3f}  mean {statistics.fmean(tt):6.3f} us/sector   "
              f"vs {engines[0][0]}: {statistics.fmean(d):+7.4f} +- {sem:6.4f}  ({100 * statistics.fmean(d) / statistics.fmean(base):+5.2f} %)")


if __name__ == "__main__":
    main()
