#!/usr/bin/env python3
"""Regenerate tests/golden/ from the reference's OWN data files and host code.

Run in the build container only (needs /root/reference).  Outputs are data --
numeric inputs and expected outputs -- never reference source text:

  ref_04abs_hh.npy      in/04abs.altb   512x512 |X|^2 of the real sector, HH   (fp32)
  ref_08pow_hh.npy      in/08pow.altb   512x512 MA-smoothed power, HH           (fp32)
  ref_09zdb.npy         in/09zdb.altb   512 Zdb           (fp64 of the 6-digit text)
  ref_10zdr.npy         in/10zdr.altb   512 Zdr
  ref_99result_cpu.npy  out/99result.cpu.out  512x2
  ref_99result_gpu.npy  out/99result.gpu.out  512x2
  ref_04abs_gpu_col256.npy  column 256 of in/04abs.altb, out/04abs.cpu.out, out/04abs.gpu.out
                        (post-shift DC bin = rounding noise, the only place the dumps differ)
  ref_cpu_bin_zdb.npy   out/cpu.bin record 0 (all 127 records are identical): fp32 Zdb of the
                        synthetic sector iq_hh[i][j] = (i, j)  (gpu_1fp.cu:295-300)
  ref_sum_out.npz       out/sum.out: 16x8 input and its per-row tree-reduction result
  ref_host_codecs.npz   vectors produced by the reference's sector.cpp / floats.c /
                        dimension.cpp compiled into oracle/_ref/libref_host.so
"""
import ctypes as C
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)


def load_txt(path):
    with open(path) as f:
        rows = [np.array(line.split(), dtype=np.float64) for line in f if line.strip()]
    return np.stack(rows)


def main():
    os.makedirs(OUT, exist_ok=True)
    a = load_txt(f"{REF}/in/04abs.altb")
    p = load_txt(f"{REF}/in/08pow.altb")
    assert a.shape == (512, 512) and p.shape == (512, 512)
    # the in/ and out/*.cpu.out copies are the same dumps
    ac = load_txt(f"{REF}/out/04abs.cpu.out")
    keep = np.arange(512) != 256   # post-shift DC bin: rounding noise, differs in all 3 dumps
    assert np.array_equal(a[:, keep], ac[:, keep])
    assert np.array_equal(p, load_txt(f"{REF}/out/08pow.cpu.out"))
    assert np.array_equal(p, load_txt(f"{REF}/out/08pow.gpu.out"))
    ag = load_txt(f"{REF}/out/04abs.gpu.out")
    diff_cols = np.unique(np.nonzero(ac != ag)[1])
    print("04abs cpu vs gpu differ only in columns", diff_cols)
    assert list(diff_cols) == [256]
    np.save(f"{OUT}/ref_04abs_hh.npy", a.astype(np.float32))
    np.save(f"{OUT}/ref_08pow_hh.npy", p.astype(np.float32))
    np.save(f"{OUT}/ref_04abs_gpu_col256.npy", np.stack([a[:, 256], ac[:, 256], ag[:, 256]]))
    np.save(f"{OUT}/ref_09zdb.npy", load_txt(f"{REF}/in/09zdb.altb")[:, 0])
    np.save(f"{OUT}/ref_10zdr.npy", load_txt(f"{REF}/in/10zdr.altb")[:, 0])
    np.save(f"{OUT}/ref_99result_cpu.npy", load_txt(f"{REF}/out/99result.cpu.out"))
    np.save(f"{OUT}/ref_99result_gpu.npy", load_txt(f"{REF}/out/99result.gpu.out"))

    b = np.fromfile(f"{REF}/out/cpu.bin", dtype=np.float32).reshape(-1, 512)
    assert b.shape[0] == 127
    assert all(np.array_equal(b[0], r, equal_nan=True) for r in b)
    np.save(f"{OUT}/ref_cpu_bin_zdb.npy", b[0])

    # out/sum.out: "in:" 16 rows of (re,im) pairs, then "out:" rows
    txt = open(f"{REF}/out/sum.out").read()
    blocks = re.split(r"^(\w+):\s*$", txt, flags=re.M)
    named = {blocks[i]: blocks[i + 1] for i in range(1, len(blocks) - 1, 2)}
    def parse(block):
        rows = []
        for line in block.strip().splitlines():
            pairs = re.findall(r"\(([-\d.e+]+),([-\d.e+]+)\)", line)
            if pairs:
                rows.append([complex(float(x), float(y)) for x, y in pairs])
        return np.array(rows, dtype=np.complex64)
    np.savez(f"{OUT}/ref_sum_out.npz", **{k: parse(v) for k, v in named.items()})
    print("sum.out blocks:", {k: parse(v).shape for k, v in named.items()})

    # host codecs through the reference's own compiled code
    from oracle import oracle
    oracle.build()
    ref = oracle.ref_host()
    assert ref is not None, "oracle/_ref/libref_host.so missing"
    rng = np.random.default_rng(20261004)
    sweeps, samples = 6, 5
    raw = rng.integers(0, 256, size=12 * sweeps * samples, dtype=np.uint8)
    raw[:12] = [0x12, 0x34, 0xFF, 0xFE, 0x80, 0x00, 0x7F, 0xFF, 0x00, 0x00, 0x00, 0x01]
    hh = np.empty(2 * sweeps * samples, np.int16); vv = np.empty_like(hh); vh = np.empty_like(hh)
    buf = (C.c_char * raw.size).from_buffer_copy(raw.tobytes())
    ref.ref_sector_from_bytes(buf, sweeps, samples,
                              hh.ctypes.data_as(C.POINTER(C.c_short)),
                              vv.ctypes.data_as(C.POINTER(C.c_short)),
                              vh.ctypes.data_as(C.POINTER(C.c_short)))
    fl = np.array([1.5, -0.0, 0.0, 3.14159274, -1e-38, 1e38, np.inf, -np.inf, 14.2001, -18.492907],
                  dtype=np.float32)
    fl = np.concatenate([fl, rng.standard_normal(54).astype(np.float32) * 100])
    ab = np.empty(4 * fl.size, np.uint8)
    ref.ref_aftoab(fl.ctypes.data_as(C.POINTER(C.c_float)), C.c_size_t(fl.size),
                   ab.ctypes.data_as(C.POINTER(C.c_ubyte)))
    back = np.empty_like(fl)
    ref.ref_abtoaf(ab.ctypes.data_as(C.POINTER(C.c_ubyte)), C.c_size_t(fl.size),
                   back.ctypes.data_as(C.POINTER(C.c_float)))
    assert np.array_equal(fl.view(np.uint32), back.view(np.uint32))
    w, h, c, d = 5, 4, 3, 3   # dimension_stub.cpp:21
    d4 = np.array([[[[ref.ref_dim4_copy_at_depth(w, h, c, d, x, y, cp, dp) for x in range(w)]
                     for y in range(h)] for cp in range(c)] for dp in range(d)], dtype=np.int32)
    d3 = np.array([[[ref.ref_dim3_at_depth(w, h, d, x, y, dp) for x in range(w)]
                    for y in range(h)] for dp in range(d)], dtype=np.int32)
    np.savez(f"{OUT}/ref_host_codecs.npz", raw=raw, sweeps=sweeps, samples=samples,
             hh=hh, vv=vv, vh=vh, floats=fl, floats_be=ab, dim4=d4, dim3=d3,
             dim_whcd=np.array([w, h, c, d]))
    for f in sorted(os.listdir(OUT)):
        print(f"{f:32s} {os.path.getsize(os.path.join(OUT, f)):9d} B")


if __name__ == "__main__":
    main()
