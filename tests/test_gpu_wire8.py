"""WRP_FLAG_WIRE_8 (round 5): the raw entries of a handle take 8 bytes per sample -- hhI hhQ vvI vvQ, big-endian int16:
the wire sample of sector.cpp:52-62 without its VH pair, which no output reads (rpv2.cu:199-213) and which the feeder drops
in the copy it makes anyway.  Everything behind the bytes must be BIT-IDENTICAL to the 12-byte entries and to
Sector::fromByteArray + the scatter of rpv2.cu:372-383 + the planar path, for every launch form: the slot path
(decode_wire<8> + two kernels), small batches, both fused launches (in-register decode), a launch that gives up and is
repeated from the raw bytes, the framed products, and the whole intermediate of the fused launches (TEE instantiations).
int16 extremes (-32768, 32767, -1, 0x7f80, 0x80ff: every byte pattern the swap could get wrong) are planted in the samples.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SHAPES = {"A": (1024, 512), "B": (2048, 128)}
EXTREMES = np.array([-32768, 32767, -1, 0x7f80, -32513, 255, -256, 1], dtype=np.int64)   # -32513 = 0x80ff as int16


@pytest.fixture(scope="module")
def wrp():
    import wrp_amd
    return wrp_amd


def _sectors(oracle, shape, count):
    m, n = SHAPES[shape]
    pool = [oracle.synthetic_sector(40 + s, m, n) for s in range(3)]
    out = []
    for k in range(count):
        s = np.roll(pool[k % 3], 5 * k, axis=-1).copy()
        # plant the extremes: a different place in every sector, both channels, I and Q
        flat = s.reshape(2, -1)
        pos = (np.arange(len(EXTREMES)) * 7919 + 104729 * k) % flat.shape[1]
        flat[0, pos] = EXTREMES + 1j * EXTREMES[::-1]
        flat[1, pos[::-1]] = EXTREMES[::-1] - 1j * EXTREMES
        flat.imag[1, pos[::-1]] = np.clip(flat.imag[1, pos[::-1]], -32768, 32767)     # -(-32768) does not exist
        out.append(s)
    return np.stack(out)


def _wire(sector, bytes_per_sample, vh_fill=0x1234):
    m, n = sector.shape[1:]
    w = np.full((m * n, bytes_per_sample // 2), vh_fill, dtype=">i2")
    for c in range(2):
        w[:, 2 * c] = sector[c].real.ravel()
        w[:, 2 * c + 1] = sector[c].imag.ravel()
    return np.frombuffer(w.tobytes(), np.uint8)


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("shape", ["A", "B"])
def test_wire8_batches_equal_the_12_byte_and_the_planar_path_bit_for_bit(wrp, oracle, shape):
    import torch
    m, n = SHAPES[shape]
    count = 19
    planar = _sectors(oracle, shape, count)
    d8 = torch.from_numpy(np.stack([_wire(p, 8) for p in planar])).cuda()
    d12 = torch.from_numpy(np.stack([_wire(p, 12) for p in planar])).cuda()
    assert d8.shape[1] == m * n * 8 and d12.shape[1] == m * n * 12
    with wrp.Engine(device=0, m=m, n=n, n_slots=1, flags=wrp.FLAG_TWO_KERNELS) as e2:
        want = e2.process_host(planar)
    # the oracle on one sector with the extremes in it (tolerance): the planted values reach the chain as written
    ref = oracle.sector(planar[3][0], planar[3][1], dtype=np.float64)
    assert np.max(np.abs(want[3][1:, 0] - ref[1:, 0]) / np.abs(ref[1:, 0])) < 1e-5 and np.max(np.abs(want[3][:, 1] - ref[:, 1])) < 2e-5
    d_out = torch.zeros(count, m // 2, 2, device="cuda")
    with wrp.Engine(device=0, m=m, n=n, n_slots=1) as e12:          # the 12-byte launch, for the record
        e12.process_batch_raw_device(d12.data_ptr(), count, d_out.data_ptr())
        e12.check()
        assert np.array_equal(_bits(d_out.cpu().numpy()), _bits(want))
    for flags, max_batch, n_sec, launches, fallbacks in (
            (wrp.FLAG_WIRE_8, 0, count, 1, 0),                                   # fused launch, in-register decode
            (wrp.FLAG_WIRE_8, 2, count, 1, 0),                                   # ... needs no workspace
            (wrp.FLAG_WIRE_8, 2, 5, 0, 0),                                       # small batch: decode_wire<8> + two kernels, in pieces
            (wrp.FLAG_WIRE_8 | wrp.FLAG_TWO_KERNELS, 4, count, 0, 0),
            (wrp.FLAG_WIRE_8 | wrp.FLAG_DEBUG_FUSED_UNDERSIZED, 8, count, 1, 1)):   # gives up: repeated from the 8-byte samples
        with wrp.Engine(device=0, m=m, n=n, n_slots=1, max_batch=max_batch, flags=flags) as e:
            d_out.fill_(float("nan"))
            e.process_batch_raw_device(d8.data_ptr(), n_sec, d_out.data_ptr())
            e.check()
            assert (e.fused_launches, e.fused_fallbacks) == (launches, fallbacks), (flags, max_batch, n_sec)
            assert np.array_equal(_bits(d_out[:n_sec].cpu().numpy()), _bits(want[:n_sec])), (shape, hex(flags), max_batch, n_sec)


@pytest.mark.parametrize("shape", ["A", "B"])
def test_wire8_slot_path_and_frames(wrp, oracle, shape):
    """wrp_pinned_raw_slot hands out m*n*8 bytes, wrp_submit_raw uploads and decodes them; results and framed products
    equal the 12-byte handle's."""
    m, n = SHAPES[shape]
    planar = _sectors(oracle, shape, 3)
    with wrp.Engine(device=0, m=m, n=n, n_slots=2, n_sectors=3, n_elevations=1) as e12, \
         wrp.Engine(device=0, m=m, n=n, n_slots=2, n_sectors=3, n_elevations=1, flags=wrp.FLAG_WIRE_8) as e8:
        assert e12.raw_slot_array(0).size == m * n * 12 and e8.raw_slot_array(0).size == m * n * 8
        for k in range(3):
            s = k % 2
            if k >= 2:
                e12.wait(s), e8.wait(s)
            e12.raw_slot_array(s)[:] = _wire(planar[k], 12)
            e8.raw_slot_array(s)[:] = _wire(planar[k], 8)
            e12.submit_raw(s, k, 0), e8.submit_raw(s, k, 0)
        for s in range(2):
            e12.wait(s), e8.wait(s)
        for k in range(3):
            assert np.array_equal(_bits(e8.result(k, 0)), _bits(e12.result(k, 0))), k
            for which in (0, 1):
                assert np.array_equal(e8.result_frame(k, 0, which), e12.result_frame(k, 0, which))
        # channels = 3: the VH plane of the decoded block is never read
    with wrp.Engine(device=0, m=m, n=n, n_slots=1, channels=3, flags=wrp.FLAG_WIRE_8) as e3, \
         wrp.Engine(device=0, m=m, n=n, n_slots=1) as e:
        e3.raw_slot_array(0)[:] = _wire(planar[1], 8)
        e3.submit_raw(0, 0, 0), e3.wait(0)
        e.raw_slot_array(0)[:] = _wire(planar[1], 12)
        e.submit_raw(0, 0, 0), e.wait(0)
        assert np.array_equal(_bits(e3.result(0, 0)), _bits(e.result(0, 0)))


@pytest.mark.parametrize("shape", ["A", "B"])
def test_wire8_framed_batch_and_whole_intermediate(wrp, oracle, shape):
    """The framed batch entry on 8-byte samples, byte for byte the 12-byte one; and the TEE instantiation of the wire8 launch:
    every [m/2][n] block that goes through the XCDs' L2 equals the two-kernel path's WRP_STAGE_MID bit for bit."""
    import torch
    m, n = SHAPES[shape]
    count = 17
    planar = _sectors(oracle, shape, count)
    d8 = torch.from_numpy(np.stack([_wire(p, 8) for p in planar])).cuda()
    d12 = torch.from_numpy(np.stack([_wire(p, 12) for p in planar])).cuda()
    words = 2 * (1 + m // 2)
    with wrp.Engine(device=0, m=m, n=n, n_slots=1, max_batch=count, flags=wrp.FLAG_WIRE_8) as e8, \
         wrp.Engine(device=0, m=m, n=n, n_slots=1, max_batch=count) as e12:
        hdr = torch.tensor([e8.frame_header(300 + k, 2) for k in range(count)], dtype=torch.int64).to(torch.int32).cuda()
        outs, frames = [], []
        for e, d in ((e8, d8), (e12, d12)):
            d_out = torch.zeros(count, m // 2, 2, device="cuda")
            d_fr = torch.zeros(count, words, dtype=torch.int32, device="cuda")
            e.process_batch_framed_device(d.data_ptr(), count, d_out.data_ptr(), d_fr.data_ptr(), hdr.data_ptr(), raw=True)
            e.check()
            assert e.fused_launches == 1 and e.fused_fallbacks == 0
            outs.append(d_out.cpu().numpy()), frames.append(d_fr.cpu().numpy())
        assert np.array_equal(_bits(outs[0]), _bits(outs[1])) and np.array_equal(frames[0], frames[1])
        d_out = torch.zeros(count, m // 2, 2, device="cuda")
        d_tee = torch.full((count, 2, m // 2, n, 2), float("nan"), device="cuda")
        rc = e8.lib.wrp_debug_fused_tee(e8.handle, C.c_void_p(d8.data_ptr()), 1, count, C.c_void_p(d_out.data_ptr()),
                                        C.c_void_p(d_tee.data_ptr()), d_tee.numel() * 4)
        assert rc == 0, e8.lib.wrp_last_hip_error(e8.handle)
        torch.cuda.synchronize()
        assert np.array_equal(_bits(d_out.cpu().numpy()), _bits(outs[0]))
        tee = d_tee.cpu().numpy().view(np.complex64)[..., 0]
        assert not np.isnan(tee.view(np.float32)).any()
        for k in (0, 8, 16):
            e12.slot_array(0)[:] = planar[k]
            e12.submit(0, 0, 0)
            e12.wait(0)
            for ch in (0, 1):
                assert np.array_equal(_bits(tee[k, ch]), _bits(e12.dump_stage(0, "mid", ch))), (shape, k, ch)
