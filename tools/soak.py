#!/usr/bin/env python3
"""Soak of the fused launches -- both shapes; planar, 12-byte and 8-byte wire samples; on the engine's stream and on a
caller's (fused launch + gated repeat per batch): N sweeps of 360 sectors each, queued back to back, the output of every
50th compared bit for bit with the two-kernel path's; fused_fallbacks must stay 0.
  python tools/soak.py [--launches 3000]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--launches", type=int, default=3000)
    args = ap.parse_args()
    import torch
    import wrp_amd
    from oracle import oracle as O
    S = 360
    for form in ("1024 x 512 planar", "1024 x 512 wire12", "1024 x 512 wire8", "2048 x 128 planar", "2048 x 128 wire12", "2048 x 128 wire8",
                 "1024 x 512 planar, caller's stream", "2048 x 128 wire8, caller's stream"):
        m, n = (2048, 128) if form.startswith("2048") else (1024, 512)
        raw = "wire" in form
        wb = 8 if "wire8" in form else 12
        flags = wrp_amd.FLAG_WIRE_8 if "wire8" in form else 0
        stream = torch.cuda.Stream() if "caller" in form else None
        pool = np.stack([O.synthetic_sector(k, m, n) for k in range(3)])
        if raw:
            w = np.zeros((3, m * n, wb // 2), dtype=">i2")
            for k in range(3):
                for c in range(2):
                    w[k, :, 2 * c] = pool[k][c].real.ravel()
                    w[k, :, 2 * c + 1] = pool[k][c].imag.ravel()
            d_pool = torch.from_numpy(np.frombuffer(w.tobytes(), np.uint8).reshape(3, m, n, wb).copy()).cuda()
        else:
            d_pool = torch.from_numpy(pool.view(np.float32)).cuda().view(3, 2, m, n, 2)
        d_in = torch.stack([torch.roll(d_pool[k % 3], shifts=k, dims=-2) for k in range(S)]).contiguous()
        d_ref = torch.zeros(S, m // 2, 2, device="cuda")
        with wrp_amd.Engine(device=0, n_slots=1, m=m, n=n, max_batch=32, flags=wrp_amd.FLAG_TWO_KERNELS | flags) as e2:
            (e2.process_batch_raw_device if raw else e2.process_batch_device)(d_in.data_ptr(), S, d_ref.data_ptr())
            e2.check()
        d_out = torch.zeros_like(d_ref)
        t0 = time.perf_counter()
        bad = 0
        with wrp_amd.Engine(device=0, n_slots=1, m=m, n=n, max_batch=32, flags=flags) as e:
            run = e.process_batch_raw_device if raw else e.process_batch_device
            for k in range(args.launches):
                run(d_in.data_ptr(), S, d_out.data_ptr(), stream=stream.cuda_stream if stream else None)
                if k % 50 == 49:
                    if stream:
                        stream.synchronize()       # stream order alone (include/wrp.h)
                    e.check()
                    bad += int(not torch.equal(d_out.view(torch.int32), d_ref.view(torch.int32)))
                    d_out.zero_()
            e.check()
            dt = time.perf_counter() - t0
            print(f"{form}: {args.launches} sweeps of {S} sectors in {dt:.1f} s ({args.launches * S / dt / 1e3:.0f} k sectors/s with the checks), "
                  f"{args.launches // 50} outputs compared with the two-kernel path: {bad} differ; fused launches {e.fused_launches}, "
                  f"fallbacks {e.fused_fallbacks}", flush=True)
            assert bad == 0 and e.fused_fallbacks == 0


if __name__ == "__main__":
    main()
