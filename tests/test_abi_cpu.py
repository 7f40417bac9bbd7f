"""CPU-only checks of the drop-in boundary: libwrp.so loads, exports every symbol that
include/wrp.h declares, and refuses bad arguments without touching a GPU."""
import ctypes as C

import pytest

import wrp_amd


def test_library_loads_and_exports_every_header_symbol():
    lib = wrp_amd.load_library()
    declared = wrp_amd.header_symbols()
    assert len(declared) >= 18
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert lib.wrp_version().decode().startswith("wrp-amd")


def test_default_config_is_the_reference_constants():
    cfg = wrp_amd.binding.default_config()
    # rpv2.cu:38-45
    assert (cfg.m, cfg.n, cfg.n_sectors, cfg.n_elevations, cfg.ma_count) == (1024, 512, 143, 9, 7)
    assert cfg.k_range_resolution == 30.0 and abs(cfg.k_calibration - 1941.05) < 1e-3
    assert cfg.channels == 2 and cfg.n_slots == 2


def test_strerror_covers_every_status():
    lib = wrp_amd.load_library()
    texts = {lib.wrp_strerror(s).decode() for s in (0, -1, -2, -3, -4, -5)}
    assert len(texts) == 6 and "ok" in texts
    assert "unknown" in lib.wrp_strerror(-99).decode()


def test_bad_arguments_are_rejected_before_any_gpu_call():
    lib = wrp_amd.load_library()
    h = C.c_void_p()
    assert lib.wrp_create(None, 0, C.byref(h)) == -1
    for field, bad in (("m", 0), ("channels", 4), ("n_slots", 0), ("ma_count", 10), ("flags", 1),
                       ("n_sectors", 0), ("max_batch", -1)):
        cfg = wrp_amd.binding.default_config(**{field: bad})
        assert lib.wrp_create(C.byref(cfg), 0, C.byref(h)) == -1, field
        assert not h.value
    # flag bits: the ones include/wrp.h names pass this check (they then fail on the missing GPU, not as INVALID); any other bit is refused
    for flags, invalid in ((wrp_amd.FLAG_WIRE_8, False), (wrp_amd.FLAG_WIRE_8 | wrp_amd.FLAG_TWO_KERNELS | 16, False),
                           (0x20000, True), (wrp_amd.FLAG_WIRE_8 | 0x100, True), (3, True)):
        cfg = wrp_amd.binding.default_config(flags=flags)
        assert (lib.wrp_create(C.byref(cfg), 0, C.byref(h)) == -1) == invalid, hex(flags)
        if h.value:                      # (a GPU box: the engine exists)
            lib.wrp_destroy(h)
            h = C.c_void_p()
    # a shape without a kernel instantiation is reported as such, not mis-run
    cfg = wrp_amd.binding.default_config(m=1000, n=512)
    assert lib.wrp_create(C.byref(cfg), 0, C.byref(h)) == -4
    # NULL handles
    assert lib.wrp_submit(None, 0, 0, 0) == -1
    assert lib.wrp_wait(None, 0) == -1
    assert lib.wrp_process_batch_device(None, None, 1, None, None) == -1
    assert lib.wrp_sector_bytes(None) == 0
    lib.wrp_destroy(None)  # no-op


def test_no_cpu_fallback_exists():
    """Without a GPU the engine must fail loudly, never compute on the host."""
    import os
    if os.path.exists("/dev/kfd"):
        pytest.skip("GPU box")
    with pytest.raises(wrp_amd.WrpError):
        wrp_amd.Engine(device=0)


def test_product_does_not_import_the_oracle():
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "weather-radar-processing_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp", ".c", ".cc")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"(from|import)\s+oracle|oracle/|liboracle|wro_", text), (dirpath, f)


def test_source_fingerprint_ignores_comments_and_layout_only():
    """bench.py reports the measured HBM traffic only when profiles/rNN/traffic.json carries the fingerprint of the code
    it runs.  The fingerprint is over the code the compiler sees: a comment or re-indentation leaves it alone, a changed
    token does not."""
    from wrp_amd.binding import _code_only
    a = "int f(int x)   // add one\n{\n    /* block\n       comment */ return x + 1;\n}\n"
    b = "int f(int x)\n{\n  return x + 1;   // reworded\n}\n"
    c = "int f(int x)\n{\n  return x + 2;\n}\n"
    assert _code_only(a) == _code_only(b)
    assert _code_only(a) != _code_only(c)
    import wrp_amd
    fp = wrp_amd.source_fingerprint()
    assert len(fp) == 16 and fp == wrp_amd.source_fingerprint()


def test_counted_wait_of_the_fused_launch_sees_its_stores_before_its_loads(tmp_path):
    """The tile members publish `stored[0]` behind `s_waitcnt vmcnt(N)`: correct only if, in PROGRAM ORDER between the
    barrier A2 and that wait, the eight slot stores come first and exactly N request(s) for the next tile follow them
    (a load hoisted above a store would let the flag overtake the data: a silent race).  Checked on the disassembly of
    every fused kernel the library is built from, with the Makefile's flags."""
    import os
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    flags = next(l for l in open(os.path.join(root, "Makefile")) if l.startswith("HIPFLAGS")).split("?=")[1].split()
    flags = [f.replace("$(ARCH)", "gfx950") for f in flags if f != "-fPIC"]
    asm = tmp_path / "engine.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "-S", "--cuda-device-only", "-o", str(asm),
                           os.path.join(root, "weather-radar-processing_amd", "csrc", "wrp_engine.hip")],
                          stderr=subprocess.DEVNULL)
    text = open(asm).read()
    checked = 0
    for name, body in re.findall(r"^(_ZN3wrp20fused_chain_\w+):.*?\n(.*?)\.amdhsa_kernel", text, flags=re.S | re.M):
        if "Lb1ELb0EEE" in name and "1024x512" in name:
            continue                      # the stamps instantiation: s_memrealtime loads sit between the stores and the wait
        ops = [l.split()[0] + (" " + l.split("vmcnt(")[1].split(")")[0] if "vmcnt(" in l else "")
               for l in body.splitlines() if re.match(r"\s+(buffer_|global_|scratch_|s_barrier|s_waitcnt vmcnt)", l)]
        waits = [i for i, o in enumerate(ops) if o in ("s_waitcnt 2", "s_waitcnt 4", "s_waitcnt 8")]
        assert waits, name
        for i in waits:
            n = int(ops[i].split()[1])
            j = max(k for k in range(i) if ops[k] == "s_barrier")
            between = [o for o in ops[j + 1:i] if not o.startswith("s_waitcnt")]
            stores = [o for o in between if "store" in o]
            if len(stores) != 8:
                continue                  # another counted wait of the kernel (not the publication of half 0)
            assert all("store" in o for o in between[:8]), (name, between)
            assert len(between) - 8 == n and all("load" in o for o in between[8:]), (name, n, between)
            checked += 1
    assert checked >= 12, checked         # 1024 x 512 planar and both wire formats, 2048 x 128 planar and both wire formats, 7 and 9 taps each
