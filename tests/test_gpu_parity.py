"""Parity of the HIP path against the oracle, through the C ABI (libwrp.so).

Tolerances (fp32 path vs the fp64 oracle), SURVEY.md §8(d2):
  stage arrays : |a-b| <= 1e-5*|b| + 1e-5*rowmax|b| per element and ||a-b||/||b|| <= 1e-6
                 (04abs: post-shift DC column n/2 excluded -- rounding noise in the reference's
                 own cpu/gpu dumps);
  Zdb / Zdr    : 1e-5 relative on Zdb, 2e-5 dB absolute on Zdr (a difference of logs, ~0; measured 9e-6);
                 gate 0 must be -inf in both.
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
def engine(wrp):
    with wrp.Engine(device=0, n_slots=3, n_sectors=8, n_elevations=2) as e:
        yield e


@pytest.fixture(scope="module")
def sectors(oracle):
    return [oracle.synthetic_sector(s) for s in range(3)]


@pytest.fixture(scope="module")
def oracle_s0(oracle, sectors):
    out = {}
    for ch in (0, 1):
        S, d = oracle.channel(sectors[0][ch], stages=True, dtype=np.float64)
        d["rowsum"] = S
        out[ch] = d
    return out


def check_final(got, want):
    assert np.isneginf(got[0, 0]) and np.isneginf(want[0, 0])
    assert np.all(np.isfinite(got[1:])) and np.all(np.isfinite(got[:, 1]))
    rel = np.max(np.abs(got[1:, 0] - want[1:, 0]) / np.abs(want[1:, 0]))
    adr = np.max(np.abs(got[:, 1] - want[:, 1]))
    assert rel < 1e-5, rel
    assert adr < 2e-5, adr


def test_single_sector_final_outputs(engine, oracle, sectors):
    """configs[1]: single sector, single stream -- Zdb/Zdr vs the fp64 oracle."""
    for s, iq in enumerate(sectors[:2]):
        engine.slot_array(0)[:] = iq
        engine.submit(0, s, 0)
        engine.wait(0)
        check_final(engine.result(s, 0), oracle.sector(iq[0], iq[1], dtype=np.float64))


@pytest.mark.parametrize("stage", ["01hamm", "02fft1", "03fft2-noshift", "03fft2", "04abs", "08pow", "rowsum"])
@pytest.mark.parametrize("ch", [0, 1])
def test_stage_by_stage(engine, sectors, oracle_s0, stage, ch):
    """out/01hamm .. 08pow equivalents, both polarisations."""
    engine.slot_array(1)[:] = sectors[0]
    engine.submit(1, 0, 1)
    engine.wait(1)
    got = engine.dump_stage(1, stage, ch)
    want = oracle_s0[ch][stage]
    if stage == "rowsum":
        assert np.max(np.abs(got - want) / np.abs(want)) < 1e-5
        return
    if stage == "02fft1":
        pass  # all m rows are compared although the chain only consumes rows < m/2
    excl = (N // 2,) if stage == "04abs" else ()
    ok, worst, l2 = stage_close(got, want, exclude_cols=excl)
    assert ok, (stage, ch, worst, l2)
    if stage in ("03fft2", "04abs", "08pow"):
        # clip: post-shift bins n-1, n-2 are exactly zero before the MA (read.cc:221-224)
        if stage != "08pow":
            assert np.all(got[:, N - 2:] == 0)


def test_full_chain_known_answer_cpu_bin(engine, oracle, golden):
    """The reference's own end-to-end answer: out/cpu.bin = fp32 Zdb of hh[i][j] = (i, j)."""
    i = np.repeat(np.arange(M)[:, None], N, 1).astype(np.float32)
    j = np.repeat(np.arange(N)[None, :], M, 0).astype(np.float32)
    a = engine.slot_array(2)
    a[0] = i + 1j * j
    a[1] = j + 1j * i
    engine.submit(2, 0, 0)
    engine.wait(2)
    got = engine.result(0, 0)[:, 0]
    ref = golden("ref_cpu_bin_zdb.npy")
    assert np.isneginf(got[0]) and np.isneginf(ref[0])
    # the reference's fp32 CPU run is itself ~4e-4 dB away from fp64 on this input
    assert np.max(np.abs(got[1:] - ref[1:])) < 2e-3
    assert oracle.rel_l2(ref, got.copy()) < 1e-5      # error.cpp's metric


def test_batch_entry_matches_slot_path_bit_for_bit(engine, sectors):
    """Same sector -> same bits whichever slot / stream / batch position ran it (§8e)."""
    ref = []
    for s, iq in enumerate(sectors):
        engine.slot_array(s % 3)[:] = iq
        engine.submit(s % 3, s, 1)
        engine.wait(s % 3)
        ref.append(engine.result(s, 1).copy())
    batch = np.stack([sectors[2], sectors[0], sectors[1], sectors[0]])
    out = engine.process_host(batch)
    for k, s in enumerate((2, 0, 1, 0)):
        assert np.array_equal(out[k].view(np.uint32), ref[s].view(np.uint32))


def test_range_pass_forms_are_bit_identical(wrp, sectors):
    """The default walking-grid range pass (next tile prefetched) and the one-tile-per-workgroup
    form, with 8- and 16-column tiles, run the same device functions: identical bits."""
    batch = np.stack([sectors[k % 3] * np.float32(1 + 0.5 * (k % 2)) for k in range(6)])
    outs = []
    for flags in (0, 8, 16, 0x400, 0x400 | 8, 0x400 | 16):
        with wrp.Engine(device=0, n_slots=1, flags=flags) as e:
            outs.append(e.process_host(batch))
    for o in outs[1:]:
        assert np.array_equal(o.view(np.uint32), outs[0].view(np.uint32))


def test_batch_larger_than_workspace_chunk(wrp, sectors):
    with wrp.Engine(device=0, n_slots=1, max_batch=2) as e:
        batch = np.stack([sectors[k % 3] for k in range(5)])
        out = e.process_host(batch)
        one = e.process_host(sectors[1][None])
        assert np.array_equal(out[1], one[0]) and np.array_equal(out[4], one[0])
        assert np.array_equal(out[0], out[3])
        assert e.process_host(batch[:0]).shape == (0, 512, 2)       # empty batch is a no-op


def test_fused_launch_is_bit_identical_to_two_kernel_path(wrp, oracle, sectors):
    """The default for batches of >= 8 sectors: ONE persistent launch, tile + row workgroups on every CU,
    XCD teams handing the intermediate over through their L2 (csrc/wrp_fused.h).  It performs the arithmetic
    of the two-kernel path element for element, so the results must agree BIT FOR BIT -- for batch sizes
    that do and do not divide evenly among the teams, when the engine is reused (every launch leaves the
    control block zeroed for the next one: there is no memset between them), and against the fp64 oracle."""
    with wrp.Engine(device=0, n_slots=1) as ef, wrp.Engine(device=0, n_slots=1, flags=wrp.FLAG_TWO_KERNELS) as e2:
        for count in (8, 19, 50):
            batch = np.stack([sectors[(3 * k + 1) % 3] * np.float32(1 + 0.25 * (k % 5)) for k in range(count)])
            a = ef.process_host(batch)
            b = e2.process_host(batch)
            assert np.all(np.isneginf(a[:, 0, 0]))
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), count
            assert np.array_equal(a.view(np.uint32), ef.process_host(batch).view(np.uint32))
            for k in (0, count - 1):
                check_final(a[k], oracle.sector(batch[k][0], batch[k][1], dtype=np.float64))
        assert ef.lib.wrp_last_hip_error(ef.handle) == b""          # no fallback happened on the way
        # fewer than 8 sectors run the two kernels by design
        assert np.array_equal(ef.process_host(batch[:3]), b[:3])


def test_fused_launch_beyond_256_tasks_per_team(wrp, sectors):
    """The hand-over flags of the fused launch are sequence numbers mod 256 (one byte per writer).  A launch over 1040
    sectors gives every XCD team 260 channel-tasks: the bytes wrap, and every sector must still come out bit for bit
    as in a small batch."""
    import torch
    S = 1040
    pool = np.stack([sectors[k % 3] * np.float32(1 + 0.25 * k) for k in range(4)])
    d_pool = torch.from_numpy(pool.view(np.float32).reshape(4, -1)).cuda()
    d_iq = d_pool[torch.arange(S, device="cuda") % 4].contiguous()
    d_out = torch.zeros(S, M // 2, 2, device="cuda")
    with wrp.Engine(device=0, n_slots=1, max_batch=S) as e, wrp.Engine(device=0, n_slots=1, flags=wrp.FLAG_TWO_KERNELS) as e2:
        e.process_batch_device(d_iq.data_ptr(), S, d_out.data_ptr())
        e.check()
        torch.cuda.synchronize()
        assert e.lib.wrp_last_hip_error(e.handle) == b""          # one fused launch, no fallback
        want = e2.process_host(pool)
    got = d_out.cpu().numpy().view(np.uint32).reshape(S // 4, 4, M // 2, 2)
    assert np.array_equal(got, np.broadcast_to(want.view(np.uint32), got.shape))


def test_fused_intermediate_equals_the_dumped_range_fft(wrp, sectors):
    """The stage dumps come from the two-kernel form; the fused launch keeps its intermediate in the XCDs' L2.
    Its hand-over slots after a launch over 8 sectors (one per XCD team) must hold, bit for bit, what went through
    them last -- half 1 (gates < m/2 with gate mod 16 >= 8) of the dumped 02fft1 stage of each team's VV channel;
    half 0 went through the same slot before it and is covered by the bit-identical final outputs above: the dumps
    describe the launch the bench times."""
    import ctypes as C
    import torch
    batch = np.stack([sectors[k % 3] * np.float32(1 + 0.125 * k) for k in range(8)])
    d_in = torch.from_numpy(batch.view(np.float32)).cuda()
    d_out = torch.zeros(8, M // 2, 2, device="cuda")
    mid = np.zeros((8, M // 4, N), np.complex64)
    rows = np.arange(M // 4)
    gates = (rows >> 3) * 16 + 8 + (rows & 7)   # the slot holds half 1 (gate mod 16 >= 8) of the team's last task
    with wrp.Engine(device=0, n_slots=1) as e:
        rc = e.lib.wrp_debug_fused_mid(e.handle, C.c_void_p(d_in.data_ptr()), 8, C.c_void_p(d_out.data_ptr()),
                                       mid.ctypes.data_as(C.c_void_p), mid.nbytes)
        assert rc == 0, e.lib.wrp_last_hip_error(e.handle)
        for k in (0, 3, 7):
            e.slot_array(0)[:] = batch[k]
            e.submit(0, 0, 0)
            e.wait(0)
            fft1 = e.dump_stage(0, "02fft1", 1)[: M // 2]
            assert np.array_equal(mid[k].view(np.uint32), fft1[gates].view(np.uint32)), k
            assert np.array_equal(fft1.view(np.uint32), e.dump_stage(0, "mid", 1).view(np.uint32))      # WRP_STAGE_MID = those rows


def test_fused_launch_through_the_device_entry_and_check(wrp, sectors):
    """wrp_process_batch_device is asynchronous: wrp_check reports a fused launch that gave up.  Here it
    must report success, on the engine's streams and on a caller's stream, back to back."""
    import torch
    count = 16
    batch = np.stack([sectors[k % 3] * np.float32(1 + k) for k in range(count)])
    d_in = torch.from_numpy(batch.view(np.float32)).cuda()
    d_a = torch.zeros(count, M // 2, 2, device="cuda")
    d_b = torch.zeros_like(d_a)
    side = torch.cuda.Stream()
    with wrp.Engine(device=0, n_slots=1) as e:
        want = e.process_host(batch)
        e.process_batch_device(d_in.data_ptr(), count, d_a.data_ptr())
        e.process_batch_device(d_in.data_ptr(), count, d_b.data_ptr(), stream=side.cuda_stream)
        e.check()
        torch.cuda.synchronize()
    assert np.array_equal(d_a.cpu().numpy().view(np.uint32), want.view(np.uint32))
    assert np.array_equal(d_b.cpu().numpy().view(np.uint32), want.view(np.uint32))


def test_fused_launch_that_cannot_form_its_teams_is_repeated(wrp, sectors):
    """A fused launch whose workgroups are not all there (here: launched with half of them; in the field: another
    kernel holding CUs) must not deliver garbage.  It says so itself, within milliseconds (the team meeting has a
    4 ms deadline in real time), and the engine repeats the batch on the two-kernel path at its next synchronisation
    point: after wrp_check the output is right, the note and the count are there, the handle stays on the two kernels
    for a while and then tries the fused launch again."""
    import time
    import torch
    count = 12
    batch = np.stack([sectors[k % 3] * np.float32(1 + k) for k in range(count)])
    d_in = torch.from_numpy(batch.view(np.float32)).cuda()
    d_out = torch.zeros(count, M // 2, 2, device="cuda")
    with wrp.Engine(device=0, n_slots=1, flags=wrp.FLAG_TWO_KERNELS) as e2:
        want = e2.process_host(batch)
    with wrp.Engine(device=0, n_slots=1, flags=wrp.FLAG_DEBUG_FUSED_UNDERSIZED) as e:
        e.process_batch_device(d_in.data_ptr(), count, d_out.data_ptr())
        t0 = time.perf_counter()
        e.check()
        dt = time.perf_counter() - t0
        assert np.array_equal(d_out.cpu().numpy().view(np.uint32), want.view(np.uint32))
        assert e.fused_fallbacks == 1
        assert b"32 tile + 32 row" in e.lib.wrp_last_hip_error(e.handle)
        assert dt < 0.25, dt                                        # it used to take a second to say "busy"
        for k in range(16):                                         # the cool-down: two kernels, no new fallback
            d_out.zero_()
            e.process_batch_device(d_in.data_ptr(), count, d_out.data_ptr())
        e.check()
        assert e.fused_fallbacks == 1
        assert np.array_equal(d_out.cpu().numpy().view(np.uint32), want.view(np.uint32))
        d_out.zero_()
        e.process_batch_device(d_in.data_ptr(), count, d_out.data_ptr())      # re-armed: tries (and, undersized, fails) again
        e.check()
        assert e.fused_fallbacks == 2
        assert np.array_equal(d_out.cpu().numpy().view(np.uint32), want.view(np.uint32))
    with wrp.Engine(device=0, n_slots=1, flags=wrp.FLAG_DEBUG_FUSED_UNDERSIZED) as e:
        assert np.array_equal(e.process_host(batch).view(np.uint32), want.view(np.uint32))
        assert b"two-kernel path" in e.lib.wrp_last_hip_error(e.handle)


def test_consecutive_fused_batches_stay_bit_identical(wrp, sectors):
    """Many fused batches back to back without a host synchronisation in between, different inputs and outputs (every
    launch leaves the control block zeroed for the next one), every one bit-identical to the two-kernel path; wrp_check
    waits for all of them."""
    import torch
    count = 24
    batches = [np.stack([sectors[(k + j) % 3] * np.float32(1 + 0.5 * ((k + j) % 4)) for k in range(count)]) for j in range(3)]
    d_in = [torch.from_numpy(b.view(np.float32)).cuda() for b in batches]
    reps = 12
    d_out = [torch.zeros(count, M // 2, 2, device="cuda") for _ in range(reps)]
    with wrp.Engine(device=0, n_slots=1, flags=wrp.FLAG_TWO_KERNELS) as e2:
        want = [e2.process_host(b) for b in batches]
    with wrp.Engine(device=0, n_slots=1) as e:
        for r in range(reps):
            e.process_batch_device(d_in[r % 3].data_ptr(), count, d_out[r].data_ptr())
        e.check()
        assert e.fused_fallbacks == 0 and e.lib.wrp_last_hip_error(e.handle) == b""
        for r in range(reps):
            assert np.array_equal(d_out[r].cpu().numpy().view(np.uint32), want[r % 3].view(np.uint32)), r


def test_wire_format_ingest_is_bit_identical_to_cpu_decode(wrp, oracle):
    """N1: raw 12 B/sample big-endian int16 upload + GPU decode == Sector::fromByteArray + the
    int16->float2 scatter of rpv2.cu:369-383 (oracle restatement, itself pinned by the reference's
    compiled sector.cpp) followed by the ordinary fp32 path.  Includes the int16 extremes."""
    rng = np.random.default_rng(11)
    raw = rng.integers(0, 256, M * N * 12, dtype=np.uint8)
    raw[:12] = [0x80, 0x00, 0x7F, 0xFF, 0xFF, 0xFF, 0x00, 0x01, 0x12, 0x34, 0xFF, 0xFE]
    hh, vv, vh = oracle.sector_from_bytes(raw, M, N)
    for channels in (2, 3):
        planar = oracle.sector_to_planar(hh, vv, vh, M, N, copies=channels)[0]      # [C][m][n]
        with wrp.Engine(device=0, n_slots=1, channels=channels) as e:
            e.raw_slot_array(0)[:] = raw
            e.submit_raw(0, 3, 1)
            e.wait(0)
            got = e.result(3, 1).copy()
            want = e.process_host(planar[None])[0]
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
            # the decoded block itself: stage 01hamm / window == the int16 values (exactly representable)
            assert e.lib.wrp_submit_raw(e.handle, 0, 999, 0) == -1


def test_vh_plane_is_carried_but_ignored(wrp, sectors):
    """channels = 3 is the reference's Dimension4(n, m, 3, streams) layout; VH feeds no output."""
    with wrp.Engine(device=0, n_slots=1, channels=3) as e3, wrp.Engine(device=0, n_slots=1) as e2:
        iq3 = np.zeros((1, 3, M, N), np.complex64)
        iq3[0, :2] = sectors[1]
        iq3[0, 2] = np.nan
        assert np.array_equal(e3.process_host(iq3), e2.process_host(sectors[1][None]))
        # the same through the fused launch (batches of >= 8 sectors): shape A3 of SURVEY 8(d), the reference-faithful
        # 12 MiB-per-sector block; the tile workgroups step over the VH plane
        count = 9
        b2 = np.stack([sectors[k % 3] * np.float32(1 + 0.5 * k) for k in range(count)])
        b3 = np.full((count, 3, M, N), np.nan, np.complex64)
        b3[:, :2] = b2
        got3 = e3.process_host(b3)
        assert np.array_equal(got3.view(np.uint32), e2.process_host(b2).view(np.uint32))
        assert e3.lib.wrp_last_hip_error(e3.handle) == b""          # the fused launch ran, no fallback


def test_special_values_compare_by_class(wrp, oracle, sectors):
    """NaN / inf compared by class.  An all-zero sector (S = 0: Zdb = -inf, Zdr = NaN), a
    silent VV channel (Zdr = +inf), one NaN sample and one inf sample (they poison their channel's column
    in the range FFT and from there every gate) -- fused launch and two-kernel path against the
    fp64 oracle; the ordinary sectors of the same batch must be unaffected."""
    base = sectors[1]
    zero = np.zeros_like(base)
    vv_silent = base.copy()
    vv_silent[1] = 0
    one_nan = base.copy()
    one_nan[0, 700, 33] = np.nan
    one_inf = base.copy()
    one_inf[1, 5, 500] = np.inf + 0j
    batch = np.stack([base, zero, vv_silent, one_nan, one_inf, base, sectors[2], zero, base])
    want = np.stack([oracle.sector(s[0], s[1], dtype=np.float64) for s in batch])
    for flags in (0, wrp.FLAG_TWO_KERNELS):
        with wrp.Engine(device=0, n_slots=1, flags=flags) as e:
            got = e.process_host(batch)
        for cls in (np.isnan, np.isposinf, np.isneginf):
            assert np.array_equal(cls(got), cls(want)), (flags, cls.__name__)
        fin = np.isfinite(want)
        assert np.max(np.abs(got[fin] - want[fin])) < 1e-4, flags          # dB
        assert np.array_equal(got[0], got[5]) and np.array_equal(got[0], got[8])
    assert np.all(np.isneginf(want[1, :, 0])) and np.all(np.isnan(want[1, :, 1]))
    assert np.all(np.isposinf(want[2, 1:, 1]))
    assert np.all(np.isnan(want[3, 1:, 0])) and np.all(np.isnan(want[4, :, 1]))


def test_scaling_property(engine, sectors):
    """Size-independent property at full size: IQ -> 2 IQ shifts Zdb by 20 log10 2, keeps Zdr."""
    base = engine.process_host(sectors[2][None])[0]
    dbl = engine.process_host((2 * sectors[2])[None])[0]
    assert np.max(np.abs(dbl[1:, 0] - base[1:, 0] - 20 * np.log10(2.0))) < 1e-4
    assert np.max(np.abs(dbl[:, 1] - base[:, 1])) < 1e-4
    # swapping the polarisations negates Zdr
    sw = engine.process_host(sectors[2][::-1][None])[0]
    assert np.max(np.abs(sw[:, 1] + base[:, 1])) < 1e-4


def test_rowsum_obeys_parseval_identity(engine, sectors, oracle_s0):
    """S[i] = n*sum_j |x_ij - mu_i|^2 - |X[n/2-2]|^2 - |X[n/2-1]|^2 on GPU dumps."""
    engine.slot_array(0)[:] = sectors[0]
    engine.submit(0, 0, 0)
    engine.wait(0)
    fft1 = engine.dump_stage(0, "02fft1", 0)[: M // 2].astype(np.complex128)
    ns = np.conj(engine.dump_stage(0, "03fft2-noshift", 0).astype(np.complex128))
    S = engine.dump_stage(0, "rowsum", 0)
    mu = fft1.mean(axis=1, keepdims=True)
    ident = N * (np.abs(fft1 - mu) ** 2).sum(axis=1) - np.abs(ns[:, N // 2 - 2]) ** 2 - np.abs(ns[:, N // 2 - 1]) ** 2
    assert np.max(np.abs(S - ident) / ident) < 1e-5


def test_rowsum_dump_ties_to_the_pow_dump(engine, sectors):
    """WRP_STAGE_ROWSUM is S as the chain forms it -- (sum of the taps) x (sum of 04abs): the DC bin of the circular moving
    average (include/wrp.h) -- not the sum of the dumped 08pow row: the two dumps must still agree to rounding,
    and S with the dumped |.|^2 row exactly as the formula says."""
    engine.slot_array(0)[:] = sectors[1]
    engine.submit(0, 0, 0)
    engine.wait(0)
    for ch in (0, 1):
        S = engine.dump_stage(0, "rowsum", ch).astype(np.float64)
        pw = engine.dump_stage(0, "08pow", ch).astype(np.float64)
        ab = engine.dump_stage(0, "04abs", ch).astype(np.float64)
        assert np.max(np.abs(S - pw.sum(axis=1)) / S) < 2e-6
        g = np.exp(-((np.arange(7) - 3) ** 2) / 2.0)
        taps_sum = float(np.sum((g / g.sum()).astype(np.float32).astype(np.float64)))
        assert np.max(np.abs(S - taps_sum * ab.sum(axis=1)) / S) < 1e-6


def test_call_order_errors(engine, sectors):
    assert engine.lib.wrp_wait(engine.handle, 0) == -5          # nothing submitted
    engine.slot_array(0)[:] = sectors[0]
    engine.submit(0, 0, 0)
    assert engine.lib.wrp_submit(engine.handle, 0, 1, 0) == -5  # slot still busy
    engine.wait(0)
    assert engine.lib.wrp_submit(engine.handle, 0, 99, 0) == -1  # sector outside the result table
    assert engine.lib.wrp_submit(engine.handle, 7, 0, 0) == -1   # no such slot


def test_products_framed_on_the_gpu_are_byte_exact(wrp, oracle, sectors):
    """the Doppler pass writes Zdb and Zdr planar and BIG-ENDIAN behind their header, the slot's D2H copy
    delivers them wire-ready and wrp_result_frame hands out the bytes where they lie.  Byte-exact against the oracle's
    framing of the same results (rpv2.cu:631-661: [sector BE16][elevation BE16][BE floats]; read_single.cc:510-520:
    [sector BE16][BE floats]) -- 1024 x 512, the tuned 2048 x 128 kernels, a generic shape, planar and wire-format ingest."""
    for m, n, flags in ((M, N, 0), (2048, 128, 0), (256, 64, 0)):
        iq = sectors[1] if (m, n) == (M, N) else oracle.synthetic_sector(3, m, n)
        with wrp.Engine(device=0, m=m, n=n, n_slots=2, n_sectors=700, n_elevations=3, flags=flags) as e:
            for slot, (sector, elev) in enumerate(((517, 2), (3, 0))):
                e.slot_array(slot)[:] = iq * np.float32(1 + slot)
                e.submit(slot, sector, elev)
            for slot, (sector, elev) in enumerate(((517, 2), (3, 0))):
                e.wait(slot)
                res = e.result(sector, elev).copy()
                for which in (0, 1):
                    for with_elev in (True, False):
                        got = e.result_frame(sector, elev, which, with_elev)
                        want = oracle.frame_result(res, sector, elev, which, with_elev)
                        assert got.tobytes() == want.tobytes(), (m, n, sector, which, with_elev)
            assert e.lib.wrp_result_frame(e.handle, 700, 0, 0, 1, None, None) == -1
    # wire-format ingest frames the same way
    hh, vv = sectors[2][0], sectors[2][1]
    w = np.zeros((M * N, 6), dtype=">i2")
    w[:, 0], w[:, 1], w[:, 2], w[:, 3] = hh.real.ravel(), hh.imag.ravel(), vv.real.ravel(), vv.imag.ravel()
    with wrp.Engine(device=0, n_slots=1, n_sectors=4, n_elevations=1) as e:
        e.raw_slot_array(0)[:] = np.frombuffer(w.tobytes(), np.uint8)
        e.submit_raw(0, 2, 0)
        e.wait(0)
        assert e.result_frame(2, 0, 1, False).tobytes() == oracle.frame_result(e.result(2, 0), 2, 0, 1, False).tobytes()


def _wire(sector, vh_fill=0):
    """[2][m][n] complex (integer-valued) -> the sector's wire bytes (sector.cpp:52-62): 12 bytes per sample, big-endian int16."""
    m, n = sector.shape[1:]
    w = np.full((m * n, 6), vh_fill, dtype=">i2")
    for c in range(2):
        w[:, 2 * c] = sector[c].real.ravel()
        w[:, 2 * c + 1] = sector[c].imag.ravel()
    return np.frombuffer(w.tobytes(), np.uint8)


def test_wire_format_batch_goes_straight_into_the_fused_launch(wrp, oracle, sectors):
    """SURVEY 8f N1 as specified: a sweep in the wire format resident on the device -> wrp_process_batch_raw_device.  For
    1024 x 512 and >= 8 sectors the tile workgroups of the persistent launch read the 12-byte samples themselves (byte swap
    + conversion in registers; 6 MiB of HBM reads per sector instead of 8, no decode pass).  Integer -> float is exact, so
    the results must be BIT-IDENTICAL to the CPU decode (oracle, pinned by the reference's compiled sector.cpp) followed by
    the planar path -- also for batches that run the two kernels (< 8 sectors), with garbage in the VH samples, for int16
    extremes, and for the 2048 x 128 shape (decoded on the GPU, then its own kernels)."""
    import torch
    rng = np.random.default_rng(5)
    count = 19
    planar = np.stack([sectors[k % 3] for k in range(count)])
    planar[7, :, :4, :8] = np.array([-32768 + 32767j, 32767 - 32768j, -1 + 0j, 0 - 1j], np.complex64)[:, None]    # extremes
    raw = np.stack([_wire(planar[k], vh_fill=int(rng.integers(-30000, 30000))) for k in range(count)])
    hh, vv, vh = oracle.sector_from_bytes(raw[7], M, N)
    assert np.array_equal(oracle.sector_to_planar(hh, vv, vh, M, N, copies=2)[0], planar[7])       # the helper == the reference decode
    d_raw = torch.from_numpy(raw).cuda()
    d_out = torch.zeros(count, M // 2, 2, device="cuda")
    with wrp.Engine(device=0, n_slots=1, flags=wrp.FLAG_TWO_KERNELS) as e2:
        want = e2.process_host(planar)
    with wrp.Engine(device=0, n_slots=1) as e:
        e.process_batch_raw_device(d_raw.data_ptr(), count, d_out.data_ptr())
        e.check()
        assert e.fused_fallbacks == 0 and e.lib.wrp_last_hip_error(e.handle) == b""
        assert np.array_equal(d_out.cpu().numpy().view(np.uint32), want.view(np.uint32))
        d_out.zero_()
        e.process_batch_raw_device(d_raw.data_ptr(), 5, d_out.data_ptr())           # the two kernels behind decode_wire
        e.check()
        assert np.array_equal(d_out[:5].cpu().numpy().view(np.uint32), want[:5].view(np.uint32))
    with wrp.Engine(device=0, n_slots=1, flags=wrp.FLAG_DEBUG_FUSED_UNDERSIZED) as e:      # a launch that gives up is repeated from the raw bytes
        d_out.zero_()
        e.process_batch_raw_device(d_raw.data_ptr(), count, d_out.data_ptr())
        e.check()
        assert e.fused_fallbacks == 1
        assert np.array_equal(d_out.cpu().numpy().view(np.uint32), want.view(np.uint32))
    m, n = 2048, 128
    pb = np.stack([oracle.synthetic_sector(s, m, n) for s in range(9)])
    d_rb = torch.from_numpy(np.stack([_wire(x, 77) for x in pb])).cuda()
    d_ob = torch.zeros(9, m // 2, 2, device="cuda")
    with wrp.Engine(device=0, m=m, n=n, n_slots=1) as e:
        e.process_batch_raw_device(d_rb.data_ptr(), 9, d_ob.data_ptr())
        e.check()
        assert np.array_equal(d_ob.cpu().numpy().view(np.uint32), e.process_host(pb).view(np.uint32))
