"""ctypes binding of include/wrp.h (libwrp.so).  No compute happens in Python."""
import ctypes as C
import os
import re

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
_LIB = None


class WrpError(RuntimeError):
    def __init__(self, status, what="", detail=""):
        self.status = status
        super().__init__(f"wrp: {what} failed with status {status}" + (f" ({detail})" if detail else ""))


class WrpConfig(C.Structure):
    """wrp_config of include/wrp.h (run-time form of rpv2.cu:38-45)."""
    _fields_ = [
        ("m", C.c_int), ("n", C.c_int), ("channels", C.c_int), ("n_slots", C.c_int),
        ("n_sectors", C.c_int), ("n_elevations", C.c_int), ("ma_count", C.c_int),
        ("k_range_resolution", C.c_float), ("k_calibration", C.c_float),
        ("max_batch", C.c_int), ("flags", C.c_int),
    ]


FLAG_ONE_TILE_PER_BLOCK, FLAG_TWO_KERNELS, FLAG_DEBUG_FUSED_UNDERSIZED, FLAG_GENERIC_KERNELS = 0x400, 0x800, 0x4000, 0x8000
FLAG_WIRE_8 = 0x10000     # the handle's raw entries take 8-byte samples (hh, vv; VH dropped by the feeder)
FUSED_MIN_SECTORS = 8

STAGE_IDS = {"01hamm": 1, "02fft1": 2, "03fft2-noshift": 3, "03fft2": 4, "04abs": 5, "08pow": 6, "rowsum": 7, "mid": 8}


def STAGE_SHAPES(m, n):
    return {
        "01hamm": ((m, n), np.complex64), "02fft1": ((m, n), np.complex64),
        "03fft2-noshift": ((m // 2, n), np.complex64), "03fft2": ((m // 2, n), np.complex64),
        "04abs": ((m // 2, n), np.float32), "08pow": ((m // 2, n), np.float32),
        "rowsum": ((m // 2,), np.float32), "mid": ((m // 2, n), np.complex64),
    }


def _code_only(text):
    """C/C++ source without comments and without the whitespace around tokens: what the compiler sees."""
    import re
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    out = []
    for line in text.split("\n"):
        # a // inside a string literal never occurs in these sources except in asm comment markers ("; ..."), which use ';'
        line = re.sub(r"//.*$", "", line)
        line = " ".join(line.split())
        if line:
            out.append(line)
    return "\n".join(out)


def source_fingerprint():
    """sha256 over the CODE libwrp.so is built from (csrc/ and include/wrp.h without comments and layout, the Makefile's
    compiler flags): ties a committed rocprof summary (profiles/rNN/traffic.json) to the code it was measured on.  Editing
    a comment does not invalidate a measurement; editing an instruction does."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.listdir(os.path.join(_PKG, "csrc")))
    for f in files:
        if f.endswith((".h", ".hip")):
            h.update(f.encode())
            h.update(_code_only(open(os.path.join(_PKG, "csrc", f)).read()).encode())
    h.update(_code_only(open(os.path.join(_ROOT, "include", "wrp.h")).read()).encode())
    for line in open(os.path.join(_ROOT, "Makefile")):
        if line.startswith("HIPFLAGS"):
            h.update(" ".join(line.split()).encode())
    return h.hexdigest()[:16]


def lib_path():
    # WRP_LIB_PATH: measurement tools load an experimental build in place of the product's (tools/ only; no test sets it)
    return os.environ.get("WRP_LIB_PATH") or os.path.join(_PKG, "lib", "libwrp.so")


def header_symbols():
    """Every function include/wrp.h declares."""
    text = open(os.path.join(_ROOT, "include", "wrp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(wrp_[a-z_0-9]+)\s*\(", text)))


def exported_symbols():
    lib = load_library()
    return [s for s in header_symbols() if hasattr(lib, s)]


def load_library():
    """dlopen libwrp.so; raise loudly when it is missing (no fallback path exists)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    p = lib_path()
    if not os.path.exists(p):
        raise WrpError(-2, "load_library", f"{p} not built -- run `make lib` / __graft_entry__.build()")
    lib = C.CDLL(p)
    vp, i, f = C.c_void_p, C.c_int, C.c_float
    fp = C.POINTER(C.c_float)
    lib.wrp_default_config.argtypes = [C.POINTER(WrpConfig)]
    lib.wrp_default_config.restype = None
    lib.wrp_create.argtypes = [C.POINTER(WrpConfig), i, C.POINTER(vp)]
    lib.wrp_destroy.argtypes = [vp]
    lib.wrp_destroy.restype = None
    lib.wrp_strerror.argtypes = [i]
    lib.wrp_strerror.restype = C.c_char_p
    lib.wrp_last_hip_error.argtypes = [vp]
    lib.wrp_last_hip_error.restype = C.c_char_p
    lib.wrp_pinned_slot.argtypes = [vp, i, C.POINTER(vp), C.POINTER(C.c_size_t)]
    lib.wrp_submit.argtypes = [vp, i, i, i]
    lib.wrp_pinned_raw_slot.argtypes = [vp, i, C.POINTER(vp), C.POINTER(C.c_size_t)]
    lib.wrp_submit_raw.argtypes = [vp, i, i, i]
    lib.wrp_debug_fused_stamps.argtypes = [vp, vp, i, vp, vp, C.c_size_t]
    lib.wrp_wait.argtypes = [vp, i]
    lib.wrp_result.argtypes = [vp, i, i, C.POINTER(fp)]
    if hasattr(lib, "wrp_result_frame"):
        lib.wrp_result_frame.argtypes = [vp, i, i, i, i, C.POINTER(C.POINTER(C.c_ubyte)), C.POINTER(C.c_size_t)]
    lib.wrp_process_device.argtypes = [vp, vp, vp, vp]
    lib.wrp_process_batch_device.argtypes = [vp, vp, i, vp, vp]
    lib.wrp_process_host.argtypes = [vp, vp, i, vp]
    if hasattr(lib, "wrp_process_batch_raw_device"):
        lib.wrp_process_batch_raw_device.argtypes = [vp, vp, i, vp, vp]
    if hasattr(lib, "wrp_process_batch_framed_device"):
        lib.wrp_process_batch_framed_device.argtypes = [vp, vp, i, vp, vp, vp, vp]
        lib.wrp_process_batch_raw_framed_device.argtypes = [vp, vp, i, vp, vp, vp, vp]
        lib.wrp_frame_header.argtypes = [i, i]
        lib.wrp_frame_header.restype = C.c_uint32
    lib.wrp_check.argtypes = [vp]
    if hasattr(lib, "wrp_fused_fallbacks"):      # absent from older builds that tools/ab.py may load beside this one
        lib.wrp_fused_fallbacks.argtypes = [vp]
    if hasattr(lib, "wrp_fused_launches"):
        lib.wrp_fused_launches.argtypes = [vp]
    lib.wrp_debug_fused_mid.argtypes = [vp, vp, i, vp, vp, C.c_size_t]
    lib.wrp_debug_fused_tee.argtypes = [vp, vp, i, i, vp, vp, C.c_size_t]
    lib.wrp_dump_stage.argtypes = [vp, i, i, i, vp]
    lib.wrp_time_batch_device.argtypes = [vp, vp, i, vp, i, fp, fp, fp]
    lib.wrp_get_config.argtypes = [vp, C.POINTER(WrpConfig)]
    for name in ("wrp_sector_bytes", "wrp_result_bytes", "wrp_algorithmic_bytes"):
        getattr(lib, name).argtypes = [vp]
        getattr(lib, name).restype = C.c_size_t
    lib.wrp_version.restype = C.c_char_p
    _LIB = lib
    return lib


def default_config(**over):
    cfg = WrpConfig()
    load_library().wrp_default_config(C.byref(cfg))
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


class Engine:
    """One handle per GPU -- thin object wrapper over wrp_create .. wrp_destroy."""

    def __init__(self, device=0, **cfg):
        self.lib = load_library()
        self.cfg = default_config(**cfg)
        self._h = C.c_void_p()
        rc = self.lib.wrp_create(C.byref(self.cfg), device, C.byref(self._h))
        if rc != 0:
            raise WrpError(rc, "wrp_create", self.lib.wrp_strerror(rc).decode())
        self.m, self.n, self.channels = self.cfg.m, self.cfg.n, self.cfg.channels
        self.gates = self.m // 2

    # -- plumbing ----------------------------------------------------------------------
    def _check(self, rc, what):
        if rc != 0:
            raise WrpError(rc, what, self.lib.wrp_strerror(rc).decode() + ": " +
                           self.lib.wrp_last_hip_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.wrp_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    @property
    def sector_bytes(self):
        return self.lib.wrp_sector_bytes(self._h)

    @property
    def algorithmic_bytes(self):
        return self.lib.wrp_algorithmic_bytes(self._h)

    # -- streaming seam (rpv2.cu do_process) --------------------------------------------
    def slot_array(self, slot):
        """numpy view [channels][m][n] complex64 of the slot's pinned staging buffer."""
        p, nbytes = C.c_void_p(), C.c_size_t()
        self._check(self.lib.wrp_pinned_slot(self._h, slot, C.byref(p), C.byref(nbytes)), "wrp_pinned_slot")
        buf = (C.c_char * nbytes.value).from_address(p.value)
        return np.frombuffer(buf, dtype=np.complex64).reshape(self.channels, self.m, self.n)

    def raw_slot_array(self, slot):
        """numpy view [m*n*12] uint8 (FLAG_WIRE_8: [m*n*8]) of the slot's pinned wire-format buffer (sector.cpp:52-62 layout)."""
        p, nbytes = C.c_void_p(), C.c_size_t()
        self._check(self.lib.wrp_pinned_raw_slot(self._h, slot, C.byref(p), C.byref(nbytes)), "wrp_pinned_raw_slot")
        buf = (C.c_char * nbytes.value).from_address(p.value)
        return np.frombuffer(buf, dtype=np.uint8)

    def submit_raw(self, slot, sector, elevation=0):
        self._check(self.lib.wrp_submit_raw(self._h, slot, sector, elevation), "wrp_submit_raw")

    def submit(self, slot, sector, elevation=0):
        self._check(self.lib.wrp_submit(self._h, slot, sector, elevation), "wrp_submit")

    def wait(self, slot):
        self._check(self.lib.wrp_wait(self._h, slot), "wrp_wait")

    def result(self, sector, elevation=0):
        p = C.POINTER(C.c_float)()
        self._check(self.lib.wrp_result(self._h, sector, elevation, C.byref(p)), "wrp_result")
        return np.ctypeslib.as_array(p, shape=(self.gates, 2))

    def result_frame(self, sector, elevation=0, which=0, with_elevation=True):
        """The product (0 = Zdb, 1 = Zdr) of (sector, elevation) as the GPU framed it for the wire: a copy of the bytes."""
        p, nbytes = C.POINTER(C.c_ubyte)(), C.c_size_t()
        self._check(self.lib.wrp_result_frame(self._h, sector, elevation, which, int(with_elevation), C.byref(p), C.byref(nbytes)),
                    "wrp_result_frame")
        return np.ctypeslib.as_array(p, shape=(nbytes.value,)).copy()

    def dump_stage(self, slot, stage, channel=0):
        shape, dt = STAGE_SHAPES(self.m, self.n)[stage]
        out = np.empty(shape, dt)
        self._check(self.lib.wrp_dump_stage(self._h, slot, STAGE_IDS[stage], channel,
                                            out.ctypes.data_as(C.c_void_p)), "wrp_dump_stage")
        return out

    # -- batch / kernel-only entries ---------------------------------------------------------
    def process_host(self, iq):
        """iq: [S][channels][m][n] complex64 on the host -> [S][m/2][2] float32."""
        iq = np.ascontiguousarray(iq, np.complex64)
        if iq.ndim == 3:
            iq = iq[None]
        assert iq.shape[1:] == (self.channels, self.m, self.n), iq.shape
        out = np.empty((iq.shape[0], self.gates, 2), np.float32)
        self._check(self.lib.wrp_process_host(self._h, iq.ctypes.data_as(C.c_void_p), iq.shape[0],
                                              out.ctypes.data_as(C.c_void_p)), "wrp_process_host")
        return out

    def process_batch_device(self, d_iq_ptr, n_sectors, d_out_ptr, stream=None):
        self._check(self.lib.wrp_process_batch_device(self._h, C.c_void_p(d_iq_ptr), n_sectors,
                                                      C.c_void_p(d_out_ptr), C.c_void_p(stream or 0)),
                    "wrp_process_batch_device")

    def process_batch_raw_device(self, d_raw_ptr, n_sectors, d_out_ptr, stream=None):
        """Wire-format batch [n_sectors][m*n][12 bytes] resident on the device (N1)."""
        self._check(self.lib.wrp_process_batch_raw_device(self._h, C.c_void_p(d_raw_ptr), n_sectors, C.c_void_p(d_out_ptr),
                                                          C.c_void_p(stream or 0)), "wrp_process_batch_raw_device")

    def process_batch_framed_device(self, d_in_ptr, n_sectors, d_out_ptr, d_frames_ptr, d_headers_ptr, stream=None, raw=False):
        """The batch entries with the products framed for the wire (N2): d_frames [S][2][1 + m/2] words, d_headers [S] words."""
        fn = self.lib.wrp_process_batch_raw_framed_device if raw else self.lib.wrp_process_batch_framed_device
        self._check(fn(self._h, C.c_void_p(d_in_ptr), n_sectors, C.c_void_p(d_out_ptr), C.c_void_p(d_frames_ptr),
                       C.c_void_p(d_headers_ptr), C.c_void_p(stream or 0)), fn.__name__)

    def frame_header(self, sector, elevation):
        return int(self.lib.wrp_frame_header(sector, elevation))

    def check(self):
        """Wait for every batch submitted so far (a fused launch that gave up is repeated on the two-kernel path here)."""
        self._check(self.lib.wrp_check(self._h), "wrp_check")

    @property
    def fused_fallbacks(self):
        return self.lib.wrp_fused_fallbacks(self._h)

    @property
    def fused_launches(self):
        return self.lib.wrp_fused_launches(self._h)

    def time_batch_device(self, d_iq_ptr, n_sectors, d_out_ptr, iters, per_kernel=False):
        """HIP-event timing on the engine's own stream -> (ms_total, ms_range, ms_doppler)."""
        t = C.c_float()
        a, b = C.c_float(), C.c_float()
        self._check(self.lib.wrp_time_batch_device(
            self._h, C.c_void_p(d_iq_ptr), n_sectors, C.c_void_p(d_out_ptr), iters, C.byref(t),
            C.byref(a) if per_kernel else None, C.byref(b) if per_kernel else None), "wrp_time_batch_device")
        return t.value, (a.value if per_kernel else None), (b.value if per_kernel else None)
