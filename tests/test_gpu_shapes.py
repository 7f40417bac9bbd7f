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


def test_shape_b_fused_launch_is_bit_identical_to_its_two_kernels(oracle):
    """configs[4] through ONE persistent launch (csrc/wrp_fused_b.h: the team protocol of the 1024 x 512 launch around
    the 16 x 16 x 8 range FFT and the 16-lane Doppler rows): the default for batches of >= 8 sectors.  Same arithmetic
    as range_pass_2048 + doppler_pass_128, element for element -> the same bits, for batch sizes that do and do not
    divide among the teams, with the reference-faithful third (VH) plane in the block, and when the engine is reused;
    and against the fp64 oracle."""
    import wrp_amd
    m, n = 2048, 128
    pool = [oracle.synthetic_sector(s, m, n) for s in range(3)]
    with wrp_amd.Engine(device=0, m=m, n=n, n_slots=1) as ef, \
            wrp_amd.Engine(device=0, m=m, n=n, n_slots=1, flags=wrp_amd.FLAG_TWO_KERNELS) as e2:
        for count in (8, 19, 40):
            batch = np.stack([pool[(3 * k + 1) % 3] * np.float32(1 + 0.25 * (k % 5)) for k in range(count)])
            a, b = ef.process_host(batch), e2.process_host(batch)
            assert np.all(np.isneginf(a[:, 0, 0]))
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), count
            assert np.array_equal(a.view(np.uint32), ef.process_host(batch).view(np.uint32))
            for k in (0, count - 1):
                want = oracle.sector(batch[k][0], batch[k][1], dtype=np.float64)
                assert np.max(np.abs(a[k, 1:, 0] - want[1:, 0]) / np.abs(want[1:, 0])) < 1e-5
                assert np.max(np.abs(a[k, :, 1] - want[:, 1])) < 2e-5
        assert ef.fused_fallbacks == 0 and ef.lib.wrp_last_hip_error(ef.handle) == b""      # the fused launch ran
        assert np.array_equal(ef.process_host(batch[:3]), b[:3])                              # < 8 sectors: the two kernels
    with wrp_amd.Engine(device=0, m=m, n=n, n_slots=1, channels=3) as e3:
        b3 = np.full((9, 3, m, n), np.nan, np.complex64)
        b3[:, :2] = batch[:9]
        assert np.array_equal(e3.process_host(b3).view(np.uint32), b[:9].view(np.uint32))
        assert e3.fused_fallbacks == 0


def test_shape_b_tuned_kernels_stage_by_stage(oracle):
    """Stage-level parity of the TUNED shape-B kernels (the dumps of test_shape_stage_by_stage_and_final for 01hamm /
    02fft1 come from the shape-generic kernels): the half-height intermediate exactly as range_pass_2048 hands it on
    (WRP_STAGE_MID = rows < m/2 of 02fft1) and the Doppler stages out of doppler_pass_128 against the fp64 oracle under
    SURVEY 8(d2)'s rule; and the fused launch's L2-resident hand-over slots, bit for bit, against that intermediate."""
    import ctypes as C
    import torch
    import wrp_amd
    m, n = 2048, 128
    batch = np.stack([oracle.synthetic_sector(s, m, n) * np.float32(1 + 0.125 * s) for s in range(8)])
    with wrp_amd.Engine(device=0, m=m, n=n, n_slots=1, n_sectors=8, n_elevations=1) as e:
        e.slot_array(0)[:] = batch[5]
        e.submit(0, 0, 0)
        e.wait(0)
        for ch in (0, 1):
            S, d = oracle.channel(batch[5][ch], stages=True, dtype=np.float64)
            ok, worst, l2 = stage_close(e.dump_stage(0, "mid", ch), d["02fft1"][: m // 2])
            assert ok, ("mid", ch, worst, l2)
            for stage in ("03fft2-noshift", "03fft2", "04abs", "08pow"):
                a = e.dump_stage(0, stage, ch)
                ok, worst, l2 = stage_close(a, d[stage], exclude_cols=(n // 2,) if stage == "04abs" else ())
                assert ok, (stage, ch, worst, l2)
                if stage in ("03fft2", "04abs"):
                    assert np.all(a[:, n - 2:] == 0)
            assert np.max(np.abs(e.dump_stage(0, "rowsum", ch) - S) / S) < 1e-5
        # the fused launch's intermediate: slot x = half 1 of sector x (8 sectors, 8 teams): [2 channels][512 rows][128]
        d_in = torch.from_numpy(batch.view(np.float32)).cuda()
        d_out = torch.zeros(8, m // 2, 2, device="cuda")
        mid = np.zeros((8, 2, 256, 16, 2, 8), np.complex64)     # [team][channel][pair row Q][tile][gate of the pair][column]
        rc = e.lib.wrp_debug_fused_mid(e.handle, C.c_void_p(d_in.data_ptr()), 8, C.c_void_p(d_out.data_ptr()),
                                       mid.ctypes.data_as(C.c_void_p), mid.nbytes)
        assert rc == 0, e.lib.wrp_last_hip_error(e.handle)
        Q = np.arange(256)

        def gate(pb):      # csrc/wrp_fused_b.h: fused_b_gate, half 1
            return 8 + (Q & 7) + 16 * (2 * ((Q >> 3) & 3) + pb + 8 * ((Q >> 5) & 1)) + 256 * (Q >> 6)
        assert sorted(np.concatenate([gate(0), gate(1)])) == [k for k in range(m // 2) if k % 16 >= 8]
        for k in (0, 3, 7):
            e.slot_array(0)[:] = batch[k]
            e.submit(0, 0, 0)
            e.wait(0)
            for ch in (0, 1):
                want = e.dump_stage(0, "mid", ch)
                for pb in (0, 1):
                    got = mid[k, ch, :, :, pb, :].reshape(256, n)
                    assert np.array_equal(got.view(np.uint32), want[gate(pb)].view(np.uint32)), (k, ch, pb)
