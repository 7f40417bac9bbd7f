#!/usr/bin/env python3
"""End-to-end streaming rate of the slot cascade (pinned H2D + chain + D2H per sector), i.e. the
PCIe-inclusive number that is NOT bench.py's `value`.  Sectors are already in the pinned slots
(no host refill), so this is the transport + GPU pipeline only.

  python tools/stream_bench.py [--slots 4] [--sectors 400] [--raw 0|1]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slots", type=int, default=4)
    ap.add_argument("--sectors", type=int, default=400)
    ap.add_argument("--raw", type=int, default=0, help="0: planar fp32 slots; 12 (or 1): 12-byte wire samples; 8: 8-byte samples (WRP_FLAG_WIRE_8)")
    ap.add_argument("--torch-first", action="store_true", help="import torch first: libwrp.so then binds to the HIP runtime bundled with torch (what bench.py runs on) instead of /opt/rocm's")
    a = ap.parse_args()
    if a.torch_first:
        import torch
        torch.cuda.init()
    import numpy as np
    import wrp_amd
    from oracle import oracle as O
    wb = 8 if a.raw == 8 else 12
    eng = wrp_amd.Engine(device=0, n_slots=a.slots, n_sectors=min(a.sectors, 4096), n_elevations=1, flags=wrp_amd.FLAG_WIRE_8 if a.raw == 8 else 0)
    iq = O.synthetic_sector(0)
    for s in range(a.slots):
        if a.raw:
            w = np.zeros((1024 * 512, wb // 2), dtype=">i2")
            w[:, 0] = iq[0].real.ravel(); w[:, 1] = iq[0].imag.ravel()
            w[:, 2] = iq[1].real.ravel(); w[:, 3] = iq[1].imag.ravel()
            eng.raw_slot_array(s)[:] = np.frombuffer(w.tobytes(), np.uint8)
        else:
            eng.slot_array(s)[:] = iq
    submit = eng.submit_raw if a.raw else eng.submit
    for rep in range(2):
        t0 = time.perf_counter()
        for k in range(a.sectors):
            s = k % a.slots
            if k >= a.slots:
                eng.wait(s)
            submit(s, k % 4096, 0)
        for s in range(min(a.slots, a.sectors)):
            eng.wait(s)
        dt = time.perf_counter() - t0
    want = O.sector(iq[0], iq[1], dtype=np.float64)
    ok = np.max(np.abs(eng.result((a.sectors - 1) % 4096, 0)[1:] - want[1:])) < 1e-3
    mb = (1024 * 512 * wb if a.raw else eng.sector_bytes) / 1e6
    hip = [ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln]
    print("HIP runtime:", sorted(set(hip)))
    print(f"{'raw int16 wire' if a.raw else 'fp32 planar'} ingest, {a.slots} slots: {a.sectors / dt:8.0f} sectors/s "
          f"end to end ({mb:.1f} MB/sector -> {a.sectors / dt * mb / 1e3:.1f} GB/s over PCIe), ok={ok}")
    eng.close()


if __name__ == "__main__":
    main()
