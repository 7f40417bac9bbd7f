"""Round 4 parity cases of the batch entries, through the C ABI (libwrp.so):

  * every ma_count the ABI accepts (1 .. 9; the reference fixes 7: rpv2.cu:45, taps read.cc:40-51 with the INTEGER
    (count - 1) / 2) through the slot path, the fused launches of both tuned shapes and the wire-format launch;
  * wire-format batches with a workspace smaller than the batch (max_batch 2);
  * batches on a CALLER's stream are right by stream order alone, also when the fused launch gives up (the gated repeat
    queued behind it): the input may be overwritten as soon as the stream has passed the batch;
  * the framed products of the batch entries (rpv2.cu:631-661, read_single.cc:510-520), byte-exact;
  * 2048 x 128 wire-format batches go straight into that shape's fused launch (in-register decode).
"""
import numpy as np
import pytest

from conftest import stage_close

pytestmark = pytest.mark.gpu

M, N = 1024, 512


@pytest.fixture(scope="module")
def wrp():
    import wrp_amd
    return wrp_amd


@pytest.fixture(scope="module")
def sectors(oracle):
    return [oracle.synthetic_sector(s) for s in range(3)]


@pytest.fixture(scope="module")
def sectors_b(oracle):
    return [oracle.synthetic_sector(10 + s, 2048, 128) for s in range(3)]


def _wire(sector, vh_fill=0):
    m, n = sector.shape[1:]
    w = np.full((m * n, 6), vh_fill, dtype=">i2")
    for c in range(2):
        w[:, 2 * c] = sector[c].real.ravel()
        w[:, 2 * c + 1] = sector[c].imag.ravel()
    return np.frombuffer(w.tobytes(), np.uint8)


def _variant(pool, k):
    """sector k of a test batch: one of the pool's sectors, its pulses rotated by k (distinct data, still int16-valued --
    the wire format cannot carry a scaled copy)"""
    return np.roll(pool[k % len(pool)], 7 * k, axis=-1)


def _final_close(got, want):
    assert np.isneginf(got[0, 0]) and np.isneginf(want[0, 0])
    rel = np.max(np.abs(got[1:, 0] - want[1:, 0]) / np.abs(want[1:, 0]))
    adr = np.max(np.abs(got[:, 1] - want[:, 1]))
    assert rel < 1e-5 and adr < 2e-5, (rel, adr)


@pytest.mark.parametrize("taps", [1, 4, 7, 9])
def test_every_ma_count_through_every_launch_form(wrp, oracle, sectors, sectors_b, taps):
    """ma_count 1, 4 (even: the integer (count - 1) / 2 of read.cc:45), 7 (the reference's) and 9 (the TAPS = 9
    instantiations): 08pow and the finals of the slot path against the fp64 oracle with the same tap count; the fused
    launch, the wire-format launch and the 2048 x 128 launch bit-identical to the two kernels and within tolerance of
    the oracle."""
    import torch
    # ---- 1024 x 512: slot path, stage 08pow + finals
    iq = sectors[1]
    S, d = oracle.channel(iq[0], n_taps=taps, stages=True, dtype=np.float64)
    want = oracle.sector(iq[0], iq[1], n_taps=taps, dtype=np.float64)
    with wrp.Engine(device=0, n_slots=1, n_sectors=2, n_elevations=1, ma_count=taps) as e:
        e.slot_array(0)[:] = iq
        e.submit(0, 1, 0)
        e.wait(0)
        _final_close(e.result(1, 0).copy(), want)
        ok, worst, l2 = stage_close(e.dump_stage(0, "08pow", 0), d["08pow"])
        assert ok, (taps, worst, l2)
        rs = e.dump_stage(0, "rowsum", 0)
        assert np.max(np.abs(rs - S) / np.abs(S)) < 1e-5
        # ---- the fused launch (>= 8 sectors), planar and wire format
        count = 9
        batch = np.stack([sectors[k % 3] for k in range(count)])
        got = e.process_host(batch)
        assert e.fused_launches == 1 and e.fused_fallbacks == 0
        d_raw = torch.from_numpy(np.stack([_wire(b, 5) for b in batch])).cuda()
        d_out = torch.zeros(count, M // 2, 2, device="cuda")
        e.process_batch_raw_device(d_raw.data_ptr(), count, d_out.data_ptr())
        e.check()
        assert e.fused_launches == 2 and e.fused_fallbacks == 0
        assert np.array_equal(d_out.cpu().numpy().view(np.uint32), got.view(np.uint32))
    with wrp.Engine(device=0, n_slots=1, ma_count=taps, flags=wrp.FLAG_TWO_KERNELS) as e2:
        assert np.array_equal(e2.process_host(batch).view(np.uint32), got.view(np.uint32))
    _final_close(got[1], want)
    # ---- 2048 x 128
    m, n = 2048, 128
    iqb = sectors_b[0]
    Sb, db = oracle.channel(iqb[1], n_taps=taps, stages=True, dtype=np.float64)
    wantb = oracle.sector(iqb[0], iqb[1], n_taps=taps, dtype=np.float64)
    bb = np.stack([sectors_b[k % 3] for k in range(8)])
    with wrp.Engine(device=0, m=m, n=n, n_slots=1, n_sectors=1, n_elevations=1, ma_count=taps) as e:
        e.slot_array(0)[:] = iqb
        e.submit(0, 0, 0)
        e.wait(0)
        _final_close(e.result(0, 0).copy(), wantb)
        ok, worst, l2 = stage_close(e.dump_stage(0, "08pow", 1), db["08pow"])
        assert ok, (taps, worst, l2)
        gotb = e.process_host(bb)
        assert e.fused_launches == 1 and e.fused_fallbacks == 0
    with wrp.Engine(device=0, m=m, n=n, n_slots=1, ma_count=taps, flags=wrp.FLAG_TWO_KERNELS) as e2:
        assert np.array_equal(e2.process_host(bb).view(np.uint32), gotb.view(np.uint32))
    _final_close(gotb[0], wantb)


def test_wire_format_batches_with_a_workspace_smaller_than_the_batch(wrp, sectors):
    """max_batch = 2 sizes the two-kernel workspace (and caps the decode workspace) at two sectors: wire-format batches of 5
    and 19 sectors must walk it in pieces -- fused launch allowed (19 sectors: the in-register decode needs no workspace)
    and forbidden -- and agree bit for bit with the planar path."""
    import torch
    count = 19
    planar = np.stack([_variant(sectors, k) for k in range(count)])
    d_raw = torch.from_numpy(np.stack([_wire(p, 9) for p in planar])).cuda()
    with wrp.Engine(device=0, n_slots=1, flags=wrp.FLAG_TWO_KERNELS) as e2:
        want = e2.process_host(planar)
    for flags in (0, wrp.FLAG_TWO_KERNELS):
        with wrp.Engine(device=0, n_slots=1, max_batch=2, flags=flags) as e:
            for n_sec in (5, 19):
                d_out = torch.zeros(n_sec, M // 2, 2, device="cuda")
                e.process_batch_raw_device(d_raw.data_ptr(), n_sec, d_out.data_ptr())
                e.check()
                assert np.array_equal(d_out.cpu().numpy().view(np.uint32), want[:n_sec].view(np.uint32)), (flags, n_sec)
            assert e.fused_fallbacks == 0
            assert e.fused_launches == (0 if flags else 1)
    # a fused launch that gives up is repeated from the raw bytes through the same small workspace
    with wrp.Engine(device=0, n_slots=1, max_batch=2, flags=wrp.FLAG_DEBUG_FUSED_UNDERSIZED) as e:
        d_out = torch.zeros(count, M // 2, 2, device="cuda")
        e.process_batch_raw_device(d_raw.data_ptr(), count, d_out.data_ptr())
        e.check()
        assert e.fused_fallbacks == 1
        assert np.array_equal(d_out.cpu().numpy().view(np.uint32), want.view(np.uint32))


def test_shape_b_wire_format_batches_take_the_fused_launch(wrp, sectors_b):
    """2048 x 128 wire-format batches go straight into that shape's persistent launch (its tile workgroups read the 12-byte
    samples: no decode pass, no workspace -- a 19-sector batch is ONE launch whatever max_batch is); a launch that gives up
    is repeated from the raw bytes through decode_wire + the two kernels, max_batch sectors at a time."""
    import torch
    m, n = 2048, 128
    count = 19
    planar = np.stack([_variant(sectors_b, k) for k in range(count)])
    d_raw = torch.from_numpy(np.stack([_wire(p, -3) for p in planar])).cuda()
    d_out = torch.zeros(count, m // 2, 2, device="cuda")
    with wrp.Engine(device=0, m=m, n=n, n_slots=1, flags=wrp.FLAG_TWO_KERNELS) as e2:
        want = e2.process_host(planar)
    for max_batch in (32, 8):
        with wrp.Engine(device=0, m=m, n=n, n_slots=1, max_batch=max_batch) as e:
            d_out.zero_()
            e.process_batch_raw_device(d_raw.data_ptr(), count, d_out.data_ptr())
            e.check()
            assert e.fused_launches == 1 and e.fused_fallbacks == 0
            assert np.array_equal(d_out.cpu().numpy().view(np.uint32), want.view(np.uint32))
    with wrp.Engine(device=0, m=m, n=n, n_slots=1, max_batch=8, flags=wrp.FLAG_DEBUG_FUSED_UNDERSIZED) as e:
        d_out.zero_()
        e.process_batch_raw_device(d_raw.data_ptr(), count, d_out.data_ptr())
        e.check()
        assert e.fused_fallbacks == 1
        assert np.array_equal(d_out.cpu().numpy().view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("shape", ["A", "A-wire", "B"])
def test_caller_stream_batches_are_right_by_stream_order_alone(wrp, sectors, sectors_b, shape):
    """include/wrp.h: a batch on a CALLER's stream is complete when the stream has passed it -- d_out is right and the
    input may be reused, without wrp_check.  Checked where it is hard: the fused launch is UNDERSIZED and gives up, the
    input buffer is overwritten right behind the stream synchronisation, a second batch with another input follows on the
    same stream, and only then does the host wait (wrp_check): results right, the failure counted, nothing recomputed
    from the overwritten input.  A healthy engine goes through the same sequence (its gated launches return at once)."""
    import torch
    m, n = (2048, 128) if shape == "B" else (M, N)
    pool = sectors_b if shape == "B" else sectors
    count = 12
    b1 = np.stack([_variant(pool, k) for k in range(count)])
    b2 = np.stack([_variant(pool, k + 5) for k in range(count)])
    with wrp.Engine(device=0, m=m, n=n, n_slots=1, flags=wrp.FLAG_TWO_KERNELS) as e2:
        want1, want2 = e2.process_host(b1), e2.process_host(b2)
    if shape == "A-wire":
        h1, h2 = np.stack([_wire(x, 1) for x in b1]), np.stack([_wire(x, 2) for x in b2])
    else:
        h1, h2 = b1.view(np.float32), b2.view(np.float32)
    side = torch.cuda.Stream()
    # (the second call finds the first launch's failure -- the stream has been synchronised -- and goes to the two kernels)
    for flags, fails in ((wrp.FLAG_DEBUG_FUSED_UNDERSIZED, 1), (0, 0)):
        d_in1, d_in2 = torch.from_numpy(h1).cuda(), torch.from_numpy(h2).cuda()
        d_o1 = torch.zeros(count, m // 2, 2, device="cuda")
        d_o2 = torch.zeros_like(d_o1)
        torch.cuda.synchronize()
        with wrp.Engine(device=0, m=m, n=n, n_slots=1, flags=flags) as e:
            call = e.process_batch_raw_device if shape == "A-wire" else e.process_batch_device
            call(d_in1.data_ptr(), count, d_o1.data_ptr(), stream=side.cuda_stream)
            side.synchronize()                       # stream order only: no wrp_check
            got1 = d_o1.cpu().numpy().copy()
            d_in1.zero_()                            # the caller recycles its sweep buffer
            torch.cuda.synchronize()
            call(d_in2.data_ptr(), count, d_o2.data_ptr(), stream=side.cuda_stream)
            side.synchronize()
            got2 = d_o2.cpu().numpy().copy()
            d_in2.zero_()
            torch.cuda.synchronize()
            e.check()                                # the host takes note; nothing is recomputed from the zeroed inputs
            assert e.fused_fallbacks == fails, (shape, flags, e.fused_fallbacks)
            assert np.array_equal(d_o1.cpu().numpy().view(np.uint32), want1.view(np.uint32))
            assert np.array_equal(d_o2.cpu().numpy().view(np.uint32), want2.view(np.uint32))
        assert np.array_equal(got1.view(np.uint32), want1.view(np.uint32)), (shape, flags)
        assert np.array_equal(got2.view(np.uint32), want2.view(np.uint32)), (shape, flags)


def test_engine_stream_batches_keep_their_input_until_check(wrp, sectors):
    """The other half of the contract: on the engine's own stream (stream = NULL) a launch that gave up is repeated by
    wrp_check FROM THE INPUT, which therefore has to stay put until then -- and does give the right answer when it does."""
    import torch
    count = 10
    b = np.stack([sectors[k % 3] for k in range(count)])
    d_in = torch.from_numpy(b.view(np.float32)).cuda()
    d_out = torch.zeros(count, M // 2, 2, device="cuda")
    with wrp.Engine(device=0, n_slots=1, flags=wrp.FLAG_TWO_KERNELS) as e2:
        want = e2.process_host(b)
    with wrp.Engine(device=0, n_slots=1, flags=wrp.FLAG_DEBUG_FUSED_UNDERSIZED) as e:
        e.process_batch_device(d_in.data_ptr(), count, d_out.data_ptr())
        e.check()
        assert e.fused_fallbacks == 1
        assert np.array_equal(d_out.cpu().numpy().view(np.uint32), want.view(np.uint32))


def _check_frames(oracle, frames, out, ids, gates):
    fw = 1 + gates
    for s, (sector, elev) in enumerate(ids):
        for which in (0, 1):
            got = frames[s, which].view(np.uint8)
            want = oracle.frame_result(out[s], sector, elev, which, True)
            assert got.tobytes() == want.tobytes(), (s, which)
            # the 2-byte header of read_single.cc:510-520 = the same bytes from offset 2 when the header word is (x, sector)
    assert frames.shape[2] == fw


@pytest.mark.parametrize("shape", ["A", "B", "generic"])
def test_batch_entries_frame_their_products_for_the_wire(wrp, oracle, sectors, sectors_b, shape):
    """SURVEY 8f N2 for batches: wrp_process_batch_framed_device / _raw_framed_device write, beside d_out, both products of
    every sector as [header word][m/2 BIG-ENDIAN floats] -- byte-exact against the oracle's framing of the same results
    (rpv2.cu:631-661) for every launch form: fused, wire-format fused, the two kernels (a small batch, the flag), a fused
    launch that gave up and was repeated (engine stream: by wrp_check; caller stream: by the gated launches); and for a shape
    without a fused kernel (256 x 64: the shape-generic kernels)."""
    import torch
    m, n = {"A": (M, N), "B": (2048, 128), "generic": (256, 64)}[shape]
    pool = sectors_b if shape == "B" else sectors if shape == "A" else [oracle.synthetic_sector(20 + s, m, n) for s in range(3)]
    fused_shape = shape != "generic"
    gates = m // 2
    count = 11
    batch = np.stack([_variant(pool, 2 * k + 1) for k in range(count)])
    ids = [((37 * k + 300) % 720, k % 5) for k in range(count)]
    d_in = torch.from_numpy(batch.view(np.float32)).cuda()
    d_raw = torch.from_numpy(np.stack([_wire(x, 4) for x in batch])).cuda()
    side = torch.cuda.Stream()
    with wrp.Engine(device=0, m=m, n=n, n_slots=1, flags=wrp.FLAG_TWO_KERNELS) as e2:
        want = e2.process_host(batch)
    cases = [(0, count, False, None), (0, count, True, None), (0, 5, False, None), (0, 5, True, None),
             (wrp.FLAG_TWO_KERNELS, count, False, None), (wrp.FLAG_DEBUG_FUSED_UNDERSIZED, count, False, None),
             (wrp.FLAG_DEBUG_FUSED_UNDERSIZED, count, True, side), (0, count, False, side)]
    for flags, n_sec, raw, stream in cases:
        with wrp.Engine(device=0, m=m, n=n, n_slots=1, flags=flags) as e:
            hdr = np.array([e.frame_header(s, el) for s, el in ids], np.uint32)
            assert hdr[0].tobytes() == bytes([ids[0][0] >> 8, ids[0][0] & 255, ids[0][1] >> 8, ids[0][1] & 255])
            d_hdr = torch.from_numpy(hdr.view(np.int32)).cuda()
            d_out = torch.zeros(n_sec, gates, 2, device="cuda")
            d_frames = torch.zeros(n_sec, 2, 1 + gates, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            e.process_batch_framed_device((d_raw if raw else d_in).data_ptr(), n_sec, d_out.data_ptr(), d_frames.data_ptr(),
                                          d_hdr.data_ptr(), stream=stream.cuda_stream if stream else None, raw=raw)
            if stream:
                stream.synchronize()
            else:
                e.check()
            out = d_out.cpu().numpy()
            frames = d_frames.cpu().numpy().view(np.uint32)
            e.check()
            assert np.array_equal(out.view(np.uint32), want[:n_sec].view(np.uint32)), (flags, n_sec, raw)
            _check_frames(oracle, frames, out, ids[:n_sec], gates)
            assert e.fused_fallbacks == (1 if flags == wrp.FLAG_DEBUG_FUSED_UNDERSIZED and fused_shape else 0)
    with wrp.Engine(device=0, m=m, n=n, n_slots=1) as e:
        assert e.lib.wrp_process_batch_framed_device(e.handle, d_in.data_ptr(), count, d_out.data_ptr(), None, None, None) == -1


@pytest.mark.parametrize("form", ["A", "A-wire", "B", "B-wire"])
def test_the_whole_fused_intermediate_equals_the_two_kernel_range_pass(wrp, sectors, sectors_b, form):
    """What the fused launches hand from their tile to their row workgroups never reaches global memory (it lives in the
    XCDs' L2), and the stage dumps come from the two-kernel path.  `wrp_debug_fused_tee` runs the launch's own
    instantiation that also copies EVERYTHING it puts through the slots -- both halves of every task, both channels, every
    sector -- to a device buffer: each [m/2][n] block must equal the two-kernel path's WRP_STAGE_MID (= rows 0 .. m/2 - 1 of
    02fft1, rpv2.cu:409-502) bit for bit, for the planar launch, the wire-format launch (its in-register decode included)
    and the 2048 x 128 launch; the finals of the same launch equal the ordinary launch's."""
    import ctypes as C
    import torch
    pool = sectors_b if form.startswith("B") else sectors
    m, n = pool[0].shape[1:]
    count = 17                                   # three sectors on team 0: its second and third reuse the slot
    batch = np.stack([_variant(pool, k) for k in range(count)])
    raw = form.endswith("-wire")
    if raw:
        d_in = torch.from_numpy(np.stack([_wire(s) for s in batch])).cuda()
    else:
        d_in = torch.from_numpy(batch.view(np.float32)).cuda()
    d_out = torch.zeros(count, m // 2, 2, device="cuda")
    d_ref = torch.zeros_like(d_out)
    d_tee = torch.full((count, 2, m // 2, n, 2), float("nan"), device="cuda")
    with wrp.Engine(device=0, n_slots=1, m=m, n=n, max_batch=count) as e:
        rc = e.lib.wrp_debug_fused_tee(e.handle, C.c_void_p(d_in.data_ptr()), int(raw), count, C.c_void_p(d_out.data_ptr()),
                                       C.c_void_p(d_tee.data_ptr()), d_tee.numel() * 4)
        assert rc == 0, e.lib.wrp_last_hip_error(e.handle)
        torch.cuda.synchronize()
        (e.process_batch_raw_device if raw else e.process_batch_device)(d_in.data_ptr(), count, d_ref.data_ptr())
        e.check()
        assert e.fused_fallbacks == 0
        assert np.array_equal(d_out.cpu().numpy().view(np.uint32), d_ref.cpu().numpy().view(np.uint32))
        tee = d_tee.cpu().numpy().view(np.complex64)[..., 0]
        assert not np.isnan(tee.view(np.float32)).any()
        for k in (0, 5, 8, 16):
            e.slot_array(0)[:] = batch[k]
            e.submit(0, 0, 0)
            e.wait(0)
            for ch in (0, 1):
                mid = e.dump_stage(0, "mid", ch)
                assert np.array_equal(tee[k, ch].view(np.uint32), mid.view(np.uint32)), (form, k, ch)


@pytest.mark.parametrize("form", ["A", "A-wire", "B", "B-wire"])
def test_fused_launches_at_ragged_batch_sizes_back_to_back(wrp, sectors, sectors_b, form):
    """The persistent launches deal sectors to 8 teams (sector s to team s mod 8): batch sizes that are no multiple of 8,
    one sector more or less than a multiple, a single task per team, more than 45 per team -- queued back to back on the
    engine's stream without a host synchronisation in between (four launches in flight share the status ring) -- must
    each give, sector for sector, the bits of the two-kernel path."""
    import torch
    pool = sectors_b if form.startswith("B") else sectors
    m, n = pool[0].shape[1:]
    total = 360
    raw = form.endswith("-wire")
    if raw:
        d_pool = torch.from_numpy(np.stack([_wire(s) for s in pool])).cuda().view(len(pool), m, n, 12)
    else:
        d_pool = torch.from_numpy(np.stack(pool).view(np.float32)).cuda().view(len(pool), 2, m, n, 2)
    # sector k: pool sector k mod 3 with its pulses rotated by k (distinct sectors, still exact int16 values)
    d_in = torch.stack([torch.roll(d_pool[k % len(pool)], shifts=k, dims=-2) for k in range(total)]).contiguous()
    sizes = [8, 9, 15, 16, 17, 23, 64, 97, 257, 360, 8]
    d_ref = torch.zeros(total, m // 2, 2, device="cuda")
    with wrp.Engine(device=0, n_slots=1, m=m, n=n, max_batch=32, flags=wrp.FLAG_TWO_KERNELS) as e2:
        (e2.process_batch_raw_device if raw else e2.process_batch_device)(d_in.data_ptr(), total, d_ref.data_ptr())
        e2.check()
    ref = d_ref.cpu().numpy().view(np.uint32)
    outs = [torch.full((s, m // 2, 2), float("nan"), device="cuda") for s in sizes]
    with wrp.Engine(device=0, n_slots=1, m=m, n=n, max_batch=32) as e:
        for s, d_out in zip(sizes, outs):
            (e.process_batch_raw_device if raw else e.process_batch_device)(d_in.data_ptr(), s, d_out.data_ptr())
        e.check()
        assert e.fused_fallbacks == 0
        assert e.fused_launches >= len(sizes)
    for s, d_out in zip(sizes, outs):
        assert np.array_equal(d_out.cpu().numpy().view(np.uint32), ref[:s]), (form, s)
