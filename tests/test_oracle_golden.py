"""Pin the CPU oracle (oracle/) against every fixture the reference still holds.

CPU-only.  Fixture provenance: tools/make_golden.py (converted from the reference's
in/*.altb, out/*.out, out/cpu.bin and from its own compiled host codecs).
"""
import numpy as np
import pytest


def test_ma_conv_reproduces_08pow_fixture(oracle, golden):
    """a7: in/04abs.altb -> in/08pow.altb (read.cc:284-301), 6-digit print precision."""
    a = golden("ref_04abs_hh.npy").astype(np.float64)
    p = golden("ref_08pow_hh.npy").astype(np.float64)
    for direct in (False, True):          # reference's FFT form and the direct 7-tap form
        q = oracle.ma_conv(a, 7, direct=direct)
        assert np.max(np.abs(q - p) / np.abs(p)) < 1e-5
    # fp32: only the direct form stays element-wise exact
    q32 = oracle.ma_conv(a.astype(np.float32), 7, direct=True)
    assert np.max(np.abs(q32 - p) / np.abs(p)) < 1e-5


def test_rowsum_reflectivity_reproduce_09zdb(oracle, golden):
    """a8+a9: in/08pow.altb -> in/09zdb.altb (read.cc:336-343)."""
    p = golden("ref_08pow_hh.npy").astype(np.float64)
    zdb_ref = golden("ref_09zdb.npy")
    s = oracle.row_sum(p)
    zdb, _ = oracle.reflectivity(s, s)
    assert np.isneginf(zdb[0]) and np.isneginf(zdb_ref[0])        # gate 0: (30*0)^2 = 0 -> -inf
    assert np.max(np.abs(zdb[1:] - zdb_ref[1:]) / np.abs(zdb_ref[1:])) < 1e-5
    # MA taps sum to one => row sums of 08pow equal row sums of 04abs
    a = golden("ref_04abs_hh.npy").astype(np.float64)
    assert np.allclose(oracle.row_sum(a), s, rtol=2e-5)


def test_99result_is_zdb_zdr(golden):
    """out/99result.{cpu,gpu}.out are the (09zdb, 10zdr) pairs (read.cc:344)."""
    for which in ("cpu", "gpu"):
        r = golden(f"ref_99result_{which}.npy")
        assert r.shape == (512, 2)
        assert np.array_equal(r[:, 0], golden("ref_09zdb.npy"))
        assert np.array_equal(r[:, 1], golden("ref_10zdr.npy"))


def test_zdr_fixture_consistent_with_formula(oracle, golden):
    """10zdr pins zdr = 10(log10 Shh - log10 Svv) only up to the lost VV power; check that the
    VV row power it implies is positive, finite and of the HH power's order (parity otherwise
    unpinned for the VV channel -- see oracle/radar_oracle.c header)."""
    s_hh = oracle.row_sum(golden("ref_08pow_hh.npy").astype(np.float64))
    zdr = golden("ref_10zdr.npy")
    s_vv = s_hh / 10 ** (zdr / 10)
    assert np.all(np.isfinite(s_vv)) and np.all(s_vv > 0)
    _, zdr2 = oracle.reflectivity(s_hh, s_vv)
    assert np.allclose(zdr2, zdr, atol=1e-9)
    assert np.all(np.abs(zdr) < 40)


@pytest.mark.parametrize("dtype,tol_db", [(np.float64, 1e-3), (np.float32, 1e-3)])
def test_full_chain_reproduces_cpu_bin(oracle, golden, dtype, tol_db):
    """a0..a9 end to end: out/cpu.bin = fp32 Zdb of the synthetic sector hh[i][j] = (i, j)
    (gpu_1fp.cu:295-300) as written by the reference's fp32 CPU program (read_single.cc:499)."""
    m, n = 1024, 512
    i = np.repeat(np.arange(m)[:, None], n, 1).astype(np.float64)
    j = np.repeat(np.arange(n)[None, :], m, 0).astype(np.float64)
    out = oracle.sector(i + 1j * j, j + 1j * i, dtype=dtype)
    ref = golden("ref_cpu_bin_zdb.npy")
    assert np.isneginf(out[0, 0]) and np.isneginf(ref[0])
    assert np.max(np.abs(out[1:, 0] - ref[1:])) < tol_db
    # error.cpp's metric (relative L2 over the finite entries)
    assert oracle.rel_l2(ref, out[:, 0].astype(np.float32)) < 1e-5


def test_fft_is_the_plain_dft(oracle):
    rng = np.random.default_rng(1)
    for n in (8, 128, 512, 1024, 2048):
        x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        assert np.allclose(oracle.fft(x, -1), np.fft.fft(x), rtol=0, atol=1e-10 * n)
        assert np.allclose(oracle.fft(x, +1), np.fft.ifft(x) * n, rtol=0, atol=1e-10 * n)
        x32 = x.astype(np.complex64)
        assert np.linalg.norm(oracle.fft(x32, -1) - np.fft.fft(x)) / np.linalg.norm(np.fft.fft(x)) < 1e-6


def test_constants(oracle):
    g = oracle.ma_coef(7)
    assert abs(g.sum() - 1) < 1e-15 and np.argmax(g) == 3 and np.allclose(g, g[::-1])
    w = oracle.hamming_coef(64, 32)
    wr = 0.53836 - 0.46164 * np.cos(2 * np.pi * np.arange(64) / 63)
    wd = 0.53836 - 0.46164 * np.cos(2 * np.pi * np.arange(32) / 31)
    c = (-1 / (16383.5 * 64 * 32 * np.sqrt(50))) / np.sqrt((wr ** 2).mean() * (wd ** 2).mean())
    assert np.allclose(w, np.outer(wr, wd) * c, rtol=1e-13)
    w32 = oracle.hamming_coef(64, 32, np.float32)
    assert np.allclose(w32, w, rtol=1e-6)
    H = oracle.ma_spectrum(512)
    gp = np.zeros(512); gp[:7] = g
    assert np.allclose(H, np.fft.fft(gp), atol=1e-14)


def test_stage_dumps_are_consistent(oracle):
    """The stage dumps obey the chain's own identities on a small seeded sector."""
    m, n = 64, 32
    rng = np.random.default_rng(7)
    x = rng.integers(-16384, 16384, (m, n)) + 1j * rng.integers(-16384, 16384, (m, n))
    S, d = oracle.channel(x, stages=True)
    w = oracle.hamming_coef(m, n)
    assert np.allclose(d["01hamm"], x * w)
    assert np.allclose(d["02fft1"], np.fft.fft(x * w, axis=0))
    r = d["02fft1"][: m // 2]
    ns = np.fft.fft(np.conj(r - r.mean(axis=1, keepdims=True)), axis=1)
    assert np.allclose(d["03fft2-noshift"], ns, atol=1e-15)
    sh = np.fft.fftshift(np.conj(ns), axes=1)
    sh[:, n - 1] = 0
    sh[:, n - 2] = 0
    assert np.allclose(d["03fft2"], sh, atol=1e-15)
    assert np.allclose(d["04abs"], np.abs(sh) ** 2)
    assert np.allclose(d["08pow"].sum(axis=1), S)
    # direct and FFT form of the MA convolution agree
    S2, d2 = oracle.channel(x, stages=True, direct=True)
    assert np.allclose(d2["08pow"], d["08pow"], rtol=1e-9, atol=1e-25)
    # Parseval identity of SURVEY §8a
    X = np.conj(ns)
    ident = n * (np.abs(r - r.mean(axis=1, keepdims=True)) ** 2).sum(axis=1) \
        - np.abs(X[:, n // 2 - 2]) ** 2 - np.abs(X[:, n // 2 - 1]) ** 2
    assert np.allclose(S, ident, rtol=1e-9)


def test_tree_sum_matches_sum_out(oracle, golden):
    z = golden("ref_sum_out.npz")
    assert np.array_equal(oracle.tree_sum_rows(z["in"]), z["out"])


def test_host_codecs_match_reference_vectors(oracle, golden):
    """Sector::fromByteArray / aftoab / Dimension3,4 restatements vs vectors produced by the
    reference's own compiled code (oracle/_ref, tools/make_golden.py)."""
    z = golden("ref_host_codecs.npz")
    hh, vv, vh = oracle.sector_from_bytes(z["raw"], int(z["sweeps"]), int(z["samples"]))
    assert np.array_equal(hh, z["hh"]) and np.array_equal(vv, z["vv"]) and np.array_equal(vh, z["vh"])
    assert list(hh[:2]) == [0x1234, -2] and vv[0] == -32768 and vv[1] == 32767   # SURVEY §2 probes
    assert np.array_equal(oracle.aftoab(z["floats"]), z["floats_be"])
    assert list(oracle.aftoab(np.float32([1.5]))) == [0x3F, 0xC0, 0x00, 0x00]
    assert np.array_equal(oracle.abtoaf(z["floats_be"]).view(np.uint32), z["floats"].view(np.uint32))
    w, h, c, d = map(int, z["dim_whcd"])
    for dp in range(d):
        for cp in range(c):
            for y in range(h):
                for x in range(w):
                    assert oracle.dim4_copy_at_depth(w, h, c, x, y, cp, dp) == z["dim4"][dp, cp, y, x]
                if cp == 0:
                    assert all(oracle.dim3_at_depth(w, h, x, y, dp) == z["dim3"][dp, y, x] for x in range(w))


def test_live_reference_host_lib_if_present(oracle):
    """When oracle/_ref/libref_host.so travelled with the tree, fuzz our restatement against it."""
    import ctypes as C
    ref = oracle.ref_host()
    if ref is None:
        pytest.skip("oracle/_ref not built (no /root/reference at build time)")
    rng = np.random.default_rng(3)
    sw, sa = 16, 8
    raw = rng.integers(0, 256, 12 * sw * sa, dtype=np.uint8)
    hh = np.empty(2 * sw * sa, np.int16); vv = np.empty_like(hh); vh = np.empty_like(hh)
    buf = (C.c_char * raw.size).from_buffer_copy(raw.tobytes())
    sp = C.POINTER(C.c_short)
    ref.ref_sector_from_bytes(buf, sw, sa, hh.ctypes.data_as(sp), vv.ctypes.data_as(sp), vh.ctypes.data_as(sp))
    o = oracle.sector_from_bytes(raw, sw, sa)
    assert all(np.array_equal(a, b) for a, b in zip(o, (hh, vv, vh)))


def test_result_framing(oracle):
    """rpv2.cu:631-644: [sector BE16][elev BE16][512 BE floats]; read_single.cc:510-517 has
    no elevation bytes."""
    z = np.arange(8, dtype=np.float32).reshape(4, 2) + 0.5
    f = oracle.frame_result(z, 0x0102, 0x0007, 0)
    assert list(f[:4]) == [1, 2, 0, 7] and len(f) == 4 + 16
    assert np.array_equal(oracle.abtoaf(f[4:]), z[:, 0])
    f2 = oracle.frame_result(z, 0x0102, 0, 1, with_elevation=False)
    assert list(f2[:2]) == [1, 2] and len(f2) == 2 + 16
    assert np.array_equal(oracle.abtoaf(f2[2:]), z[:, 1])
