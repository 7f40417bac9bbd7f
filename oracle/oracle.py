"""ctypes front-end of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from the product package.  All functions
follow oracle/radar_oracle.c, which cites the reference lines it restates.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

c_double_p = C.POINTER(C.c_double)
c_float_p = C.POINTER(C.c_float)


def build(force=False):
    """make -C oracle (and the reference host lib when /root/reference exists)."""
    so = os.path.join(_HERE, "liboracle.so")
    if force or not os.path.exists(so) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(so)
            for f in ("radar_oracle.c", "radar_oracle_impl.inc")):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.wro_btof.restype = C.c_float
        _LIB.wro_rel_l2.restype = C.c_float
        _LIB.wro_frame_result.restype = C.c_size_t
    return _LIB


def ref_host():
    """The reference's own sector/floats/dimension code (oracle/_ref), or None."""
    global _REF
    if _REF is None:
        p = os.path.join(_HERE, "_ref", "libref_host.so")
        if not os.path.exists(p):
            return None
        _REF = C.CDLL(p)
        _REF.ref_btof.restype = C.c_float
    return _REF


def _sfx(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float64:
        return "f64", C.c_double, np.complex128
    if dtype == np.float32:
        return "f32", C.c_float, np.complex64
    raise TypeError(dtype)


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct)) if a is not None else None


def hamming_coef(m, n, dtype=np.float64):
    s, ct, _ = _sfx(dtype)
    out = np.empty((m, n), dtype=dtype)
    getattr(lib(), "wro_hamming_coef_" + s)(m, n, _p(out, ct))
    return out


def ma_coef(n_taps=7, dtype=np.float64):
    s, ct, _ = _sfx(dtype)
    out = np.empty(n_taps, dtype=dtype)
    getattr(lib(), "wro_ma_coef_" + s)(n_taps, _p(out, ct))
    return out


def ma_spectrum(n, n_taps=7, dtype=np.float64):
    s, ct, cdt = _sfx(dtype)
    out = np.empty(n, dtype=cdt)
    getattr(lib(), "wro_ma_spectrum_" + s)(n, n_taps, _p(out, ct))
    return out


def fft(x, sign=-1):
    """Unnormalised DFT with exp(sign*2*pi*i*j*k/n), like fftw_execute."""
    x = np.array(x, copy=True)
    s, ct, cdt = _sfx(x.real.dtype)
    x = np.ascontiguousarray(x, dtype=cdt)
    getattr(lib(), "wro_fft_" + s)(x.shape[-1], sign, _p(x, ct))
    return x


def ma_conv(abs2, n_taps=7, direct=False):
    s, ct, _ = _sfx(abs2.dtype)
    a = np.ascontiguousarray(abs2)
    out = np.empty_like(a)
    fn = "wro_ma_conv_direct_" if direct else "wro_ma_conv_fft_"
    getattr(lib(), fn + s)(a.shape[0], a.shape[1], n_taps, _p(a, ct), _p(out, ct))
    return out


def row_sum(pw):
    s, ct, _ = _sfx(pw.dtype)
    a = np.ascontiguousarray(pw)
    out = np.empty(a.shape[0], dtype=a.dtype)
    getattr(lib(), "wro_row_sum_" + s)(a.shape[0], a.shape[1], _p(a, ct), _p(out, ct))
    return out


def reflectivity(s_hh, s_vv, k_rangeres=30.0, k_calib=1941.05):
    s, ct, _ = _sfx(s_hh.dtype)
    a = np.ascontiguousarray(s_hh)
    b = np.ascontiguousarray(s_vv)
    zdb = np.empty_like(a)
    zdr = np.empty_like(a)
    with np.errstate(all="ignore"):
        getattr(lib(), "wro_reflectivity_" + s)(a.shape[0], _p(a, ct), _p(b, ct),
                                                ct(k_rangeres), ct(k_calib), _p(zdb, ct), _p(zdr, ct))
    return zdb, zdr


STAGES = ("01hamm", "02fft1", "03fft2-noshift", "03fft2", "04abs", "08pow")


def channel(iq, n_taps=7, direct=False, stages=False, dtype=np.float64, coef=None):
    """Run one channel (m x n complex) through a2..a8.  Returns (S[m/2], dumps)."""
    s, ct, cdt = _sfx(dtype)
    m, n = iq.shape
    x = np.ascontiguousarray(iq, dtype=cdt).copy()
    if coef is None:
        coef = hamming_coef(m, n, dtype)
    S = np.empty(m // 2, dtype=dtype)
    d = {}
    if stages:
        d["01hamm"] = np.empty((m, n), cdt)
        d["02fft1"] = np.empty((m, n), cdt)
        d["03fft2-noshift"] = np.empty((m // 2, n), cdt)
        d["03fft2"] = np.empty((m // 2, n), cdt)
        d["04abs"] = np.empty((m // 2, n), dtype)
        d["08pow"] = np.empty((m // 2, n), dtype)
    getattr(lib(), "wro_channel_" + s)(
        m, n, n_taps, int(direct), _p(coef, ct), _p(x, ct), _p(S, ct),
        *[_p(d.get(k), ct) for k in STAGES])
    return S, d


def sector(iq_hh, iq_vv, n_taps=7, direct=False, k_rangeres=30.0, k_calib=1941.05,
           dtype=np.float64, coef=None):
    """One sector -> zdb_zdr[m/2][2] (layout of rpv2.cu:211-212)."""
    s, ct, cdt = _sfx(dtype)
    m, n = iq_hh.shape
    a = np.ascontiguousarray(iq_hh, dtype=cdt).copy()
    b = np.ascontiguousarray(iq_vv, dtype=cdt).copy()
    if coef is None:
        coef = hamming_coef(m, n, dtype)
    out = np.empty((m // 2, 2), dtype=dtype)
    with np.errstate(all="ignore"):
        getattr(lib(), "wro_sector_" + s)(m, n, n_taps, int(direct), ct(k_rangeres), ct(k_calib),
                                          _p(coef, ct), _p(a, ct), _p(b, ct), _p(out, ct))
    return out


# ---- host codecs -----------------------------------------------------------

def sector_from_bytes(buf, sweeps, samples):
    raw = np.frombuffer(bytes(buf), dtype=np.uint8)
    assert raw.size == 12 * sweeps * samples
    hh = np.empty(2 * sweeps * samples, np.int16)
    vv = np.empty_like(hh)
    vh = np.empty_like(hh)
    lib().wro_sector_from_bytes(_p(raw, C.c_ubyte), sweeps, samples,
                                _p(hh, C.c_short), _p(vv, C.c_short), _p(vh, C.c_short))
    return hh, vv, vh


def sector_to_planar(hh, vv, vh, m, n, copies=3, slots=1, slot=0):
    out = np.zeros((slots, copies, m, n), np.complex64)
    lib().wro_sector_to_planar(_p(hh, C.c_short), _p(vv, C.c_short), _p(vh, C.c_short),
                               m, n, copies, slot, _p(out, C.c_float))
    return out


def aftoab(af):
    af = np.ascontiguousarray(af, np.float32)
    out = np.empty(4 * af.size, np.uint8)
    lib().wro_aftoab(_p(af, C.c_float), C.c_size_t(af.size), _p(out, C.c_ubyte))
    return out


def abtoaf(ab):
    ab = np.ascontiguousarray(ab, np.uint8)
    out = np.empty(ab.size // 4, np.float32)
    lib().wro_abtoaf(_p(ab, C.c_ubyte), C.c_size_t(out.size), _p(out, C.c_float))
    return out


def dim4_copy_at_depth(w, h, copies, x, y, copy, depth):
    return lib().wro_dim4_copy_at_depth(w, h, copies, x, y, copy, depth)


def dim3_at_depth(w, h, x, y, depth):
    return lib().wro_dim3_at_depth(w, h, x, y, depth)


def frame_result(zdb_zdr, sector_id, elevation, which, with_elevation=True):
    z = np.ascontiguousarray(zdb_zdr, np.float32)
    gates = z.shape[0]
    out = np.empty(4 * gates + 4, np.uint8)
    k = lib().wro_frame_result(_p(z, C.c_float), gates, sector_id, elevation, which,
                               int(with_elevation), _p(out, C.c_ubyte))
    return out[:k].copy()


def rel_l2(cpu, gpu):
    a = np.ascontiguousarray(cpu, np.float32)
    b = np.ascontiguousarray(gpu, np.float32)
    return float(lib().wro_rel_l2(_p(a, C.c_float), _p(b, C.c_float), min(a.size, b.size)))


def tree_sum_rows(x):
    x = np.ascontiguousarray(x, np.complex64)
    out = np.empty_like(x)
    lib().wro_tree_sum_rows(x.shape[0], x.shape[1], _p(x, C.c_float), _p(out, C.c_float))
    return out


# ---- synthetic workload --------------------------------------

def synthetic_sector(s, m=1024, n=512, channels=2):
    """Deterministic sector `s`: int16 noise in +-16384 plus three tone gates.

    Returns complex64 [channels][m][n] holding integer-valued I/Q like the wire
    format delivers (the tone is rounded to int16 too)."""
    rng = np.random.default_rng(0x5EED0000 + s)
    iq = rng.integers(-16384, 16384, size=(channels, m, n, 2), dtype=np.int16).astype(np.float64)
    i = np.arange(m)[:, None]
    j = np.arange(n)[None, :]
    for ch in range(channels):
        for g, amp in ((64, 8000.0), (200, 8000.0 * (0.5 + 0.25 * ch)), (333, 4000.0)):
            tone = amp * np.exp(2j * np.pi * (g * i / m + 0.1 * j))
            iq[ch, :, :, 0] += tone.real
            iq[ch, :, :, 1] += tone.imag
    iq = np.clip(np.rint(iq), -32768, 32767)
    return (iq[..., 0] + 1j * iq[..., 1]).astype(np.complex64)
