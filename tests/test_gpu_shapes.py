"""GPU parity for the other sector shapes of BASELINE.json's configs (config 5: 2048 range cells x
128 pulses) and a small shape, through the same C ABI (generic kernels, wrp_generic.h):
stage by stage and final Zdb/Zdr against the fp64 oracle, same tolerances as test_gpu_parity."""
import numpy as np
import pytest

from conftest import stage_close

pytestmark = pytest.mark.gpu

SHAPES = [(2048, 128), (64, 32), (256, 1024)]


@pytest.mark.parametrize("m,n", SHAPES)
def test_shape_stage_by_stage_and_final(oracle, m, n):
    import wrp_amd
    iq = oracle.synthetic_sector(5, m, n)
    with wrp_amd.Engine(device=0, m=m, n=n, n_slots=1, n_sectors=2, n_elevations=1) as e:
        e.slot_array(0)[:] = iq
        e.submit(0, 1, 0)
        e.wait(0)
        got = e.result(1, 0).copy()
        want = oracle.sector(iq[0], iq[1], dtype=np.float64)
        assert got.shape == (m // 2, 2)
        assert np.isneginf(got[0, 0]) and np.isneginf(want[0, 0])
        assert np.max(np.abs(got[1:, 0] - want[1:, 0]) / np.abs(want[1:, 0])) < 1e-5
        assert np.max(np.abs(got[:, 1] - want[:, 1])) < 2e-5
        for ch in (0, 1):
            S, d = oracle.channel(iq[ch], stages=True, dtype=np.float64)
            for stage in ("01hamm", "02fft1", "03fft2-noshift", "03fft2", "04abs", "08pow"):
                a = e.dump_stage(0, stage, ch)
                excl = (n // 2,) if stage == "04abs" else ()
                ok, worst, l2 = stage_close(a, d[stage], exclude_cols=excl)
                assert ok, (m, n, stage, ch, worst, l2)
            assert np.max(np.abs(e.dump_stage(0, "rowsum", ch) - S) / S) < 1e-5
        # batch entry and wire ingest work for this shape too
        out = e.process_host(np.stack([iq, iq]))
        assert np.array_equal(out[0], got) and np.array_equal(out[1], got)


def test_unsupported_shapes_are_refused():
    import ctypes as C
    import wrp_amd
    lib = wrp_amd.load_library()
    h = C.c_void_p()
    for m, n in ((4096, 512), (1024, 2048), (32, 32), (1024, 16), (1000, 512)):
        cfg = wrp_amd.binding.default_config(m=m, n=n)
        assert lib.wrp_create(C.byref(cfg), 0, C.byref(h)) == -4, (m, n)


def test_shape_b_tuned_kernels_against_generic_and_oracle(oracle):
    """configs[4]'s 2048 x 128 has tuned kernels (csrc/wrp_shape_b.h: 16 x 16 x 8 range FFT through a grouped LDS
    image, 16 lanes per Doppler row); the shape-generic radix-2 kernels produce its stage dumps.  Final outputs
    of both against the fp64 oracle and against each other, for a batch that keeps the walking grid busy for
    several tiles per workgroup and for distinct sectors (a tile / gate mix-up would show)."""
    import wrp_amd
    m, n = 2048, 128
    batch = np.stack([oracle.synthetic_sector(s, m, n) * np.float32(1 + 0.5 * (s % 3)) for s in range(40)])
    with wrp_amd.Engine(device=0, m=m, n=n, n_slots=1) as et, \
            wrp_amd.Engine(device=0, m=m, n=n, n_slots=1, flags=wrp_amd.FLAG_GENERIC_KERNELS) as eg:
        a, b = et.process_host(batch), eg.process_host(batch)
    assert not np.array_equal(a, b)                                  # really two implementations
    assert np.all(np.isneginf(a[:, 0, 0])) and np.all(np.isneginf(b[:, 0, 0]))
    assert np.max(np.abs(a[:, 1:, 0] - b[:, 1:, 0])) < 2e-5 and np.max(np.abs(a[:, :, 1] - b[:, :, 1])) < 2e-5      # dB
    for s in (0, 17, 39):
        want = oracle.sector(batch[s][0], batch[s][1], dtype=np.float64)
        assert np.max(np.abs(a[s, 1:, 0] - want[1:, 0]) / np.abs(want[1:, 0])) < 1e-5
        assert np.max(np.abs(a[s, :, 1] - want[:, 1])) < 2e-5
