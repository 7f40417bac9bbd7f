#!/usr/bin/env python3
"""Interleaved A/B timing of engine variants in ONE process (cdna guide §5.4 rule 24).

  python tools/tune.py --variants "tcols=16,mb=24;tcols=8,mb=24" --sectors 360 --rounds 5

Each variant is a comma list of key=value: fused (1 default | 0 = two kernels), tcols (two-kernel range-pass tile, 8|16), onetile, mb (max_batch).
Reports median / min us per sector for total, range pass and Doppler pass.
"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="fused=1;fused=0")
    ap.add_argument("--sectors", type=int, default=360)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()

    import numpy as np
    import torch
    import wrp_amd
    from oracle import oracle as O

    dev = torch.device("cuda", 0)
    S = args.sectors
    pool = np.stack([O.synthetic_sector(k) for k in range(4)])
    d_pool = torch.from_numpy(pool.view(np.float32).reshape(4, -1)).to(dev)
    d_iq = d_pool[torch.arange(S, device=dev) % 4].contiguous()
    d_out = torch.empty((S, 512, 2), dtype=torch.float32, device=dev)
    want = O.sector(pool[1][0], pool[1][1], dtype=np.float64)

    engines = []
    for spec in args.variants.split(";"):
        kv = dict(x.split("=") for x in spec.split(","))
        cfg = dict(n_slots=1, n_sectors=1, n_elevations=1)
        cfg["flags"] = (int(kv.get("tcols", 0)) | (0 if int(kv.get("fused", 1)) else 0x800) |
                        (0x400 if int(kv.get("onetile", 0)) else 0))
        if "mb" in kv:
            cfg["max_batch"] = int(kv["mb"])
        e = wrp_amd.Engine(device=0, **cfg)
        e.process_batch_device(d_iq.data_ptr(), S, d_out.data_ptr())
        torch.cuda.synchronize()
        got = d_out[1].cpu().numpy()
        ok = np.max(np.abs(got[1:] - want[1:])) < 1e-3
        engines.append((spec, e, ok, [], [], []))
    for _ in range(args.rounds):
        for spec, e, ok, tt, tr, td in engines:
            t, r, d = e.time_batch_device(d_iq.data_ptr(), S, d_out.data_ptr(), args.iters, per_kernel=True)
            k = 1e3 / (args.iters * S)
            tt.append(t * k)
            tr.append(r * k)
            td.append(d * k)
    for spec, e, ok, tt, tr, td in engines:
        print(f"{spec:28s} ok={ok}  total {statistics.median(tt):6.3f} (min {min(tt):6.3f})  "
              f"range {statistics.median(tr):6.3f} (min {min(tr):6.3f})  "
              f"doppler {statistics.median(td):6.3f} (min {min(td):6.3f})  us/sector  "
              f"-> {8392704 / statistics.median(tt) / 1e3:7.1f} GB/s algorithmic")
        e.close()


if __name__ == "__main__":
    main()
