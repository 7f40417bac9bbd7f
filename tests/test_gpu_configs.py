"""BASELINE.json configs[2] and configs[3] on one GPU, through the C ABI.

configs[2]: one elevation sweep of 360 sectors through the 4-slot cascade (pinned H2D overlapped with the
            kernels), fp32 planar and wire-format ingest, slots waited for out of submission order.
configs[3]: a 10 x 360 volume scan sharded by sector index over 8 GPUs -- ranks 0 and 7 of
            sharding.volume_plan(10, 360, r, 8) run here, on the one GPU there is; their rows of the result
            table must carry the same bits as the batch entry (which runs the fused launch): a sector's
            result does not depend on the GPU, slot, stream or launch form that produced it.
Pattern of the slot loop: gpu_1fp_streamcasc.cu:527-737 (without its unsynchronised host read at :695-697).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
M, N = 1024, 512
SLOTS = 4


@pytest.fixture(scope="module")
def wrp():
    import wrp_amd
    return wrp_amd


@pytest.fixture(scope="module")
def pool(oracle):
    return [oracle.synthetic_sector(s) for s in range(6)]


def sector_data(pool, e, s):
    """A distinct, exactly reproducible sector for (elevation, sector): a pool member scaled by a power-of-two
    fraction (exact in fp32, so the planar block and the oracle see the same numbers)."""
    return pool[(7 * e + s) % len(pool)] * np.float32(1 + ((s * 13 + e * 5) % 64) / 64)


def wire_bytes(iq):
    s = np.zeros((M * N, 6), dtype=">i2")
    s[:, 0] = iq[0].real.ravel(); s[:, 1] = iq[0].imag.ravel()
    s[:, 2] = iq[1].real.ravel(); s[:, 3] = iq[1].imag.ravel()
    return np.frombuffer(s.tobytes(), np.uint8)


def run_cascade(eng, items, fill, submit, order):
    """items: [(elev, sector, payload)]; SLOTS at a time are in flight; `order` permutes the waits of a round."""
    for base in range(0, len(items), SLOTS):
        group = items[base:base + SLOTS]
        for slot, (e, s, payload) in enumerate(group):
            fill(slot, payload)
            submit(slot, s, e)
        for slot in [o for o in order if o < len(group)]:
            eng.wait(slot)


def test_sweep_of_360_sectors_through_four_slots(wrp, oracle, pool):
    S = 360
    with wrp.Engine(device=0, n_slots=SLOTS, n_sectors=S, n_elevations=2) as e:
        items = [(0, s, sector_data(pool, 0, s)) for s in range(S)]
        run_cascade(e, items, lambda slot, iq: e.slot_array(slot).__setitem__(slice(None), iq), e.submit, (2, 0, 3, 1))
        planar = np.stack([e.result(s, 0).copy() for s in range(S)])
        # the same sweep as it arrives on the wire (integer samples only: the pool's own values, unscaled)
        wire_items = [(1, s, wire_bytes(pool[s % len(pool)])) for s in range(S)]
        run_cascade(e, wire_items, lambda slot, raw: e.raw_slot_array(slot).__setitem__(slice(None), raw), e.submit_raw,
                    (3, 1, 0, 2))
        wired = np.stack([e.result(s, 1).copy() for s in range(S)])
        # bit-equality with the batch entry (fused launch), 45 sectors at a time
        for c0 in range(0, S, 45):
            batch = np.stack([items[s][2] for s in range(c0, c0 + 45)])
            assert np.array_equal(e.process_host(batch).view(np.uint32), planar[c0:c0 + 45].view(np.uint32)), c0
        assert e.lib.wrp_last_hip_error(e.handle) == b""
    for s in (0, 1, 179, 359):
        want = oracle.sector(items[s][2][0], items[s][2][1], dtype=np.float64)
        assert np.isneginf(planar[s, 0, 0])
        assert np.max(np.abs(planar[s, 1:, 0] - want[1:, 0]) / np.abs(want[1:, 0])) < 1e-5
        assert np.max(np.abs(planar[s, :, 1] - want[:, 1])) < 2e-5
    base = {k: None for k in range(len(pool))}
    for s in range(S):          # wire ingest: every sector equals the planar result of its pool member
        k = s % len(pool)
        if base[k] is None:
            base[k] = wired[s]
        assert np.array_equal(wired[s].view(np.uint32), base[k].view(np.uint32)), s


@pytest.mark.parametrize("rank", [0, 7])
def test_volume_scan_shard_of_one_rank(wrp, pool, rank):
    from wrp_amd import sharding
    E, S, G = 10, 360, 8
    plan = sharding.volume_plan(E, S, rank, G)
    assert len(plan) == E * (S // G) and all(s % G == rank for _, s in plan)
    with wrp.Engine(device=0, n_slots=SLOTS, n_sectors=S, n_elevations=E) as e:
        items = [(el, s, sector_data(pool, el, s)) for el, s in plan]
        run_cascade(e, items, lambda slot, iq: e.slot_array(slot).__setitem__(slice(None), iq), e.submit, (1, 3, 0, 2))
        got = np.stack([e.result(s, el).copy() for el, s in plan])
        # rows of the table this rank does not own stay untouched
        other = (rank + 1) % G
        assert not np.any(e.result(other, 0))
        for c0 in range(0, len(items), 90):
            batch = np.stack([it[2] for it in items[c0:c0 + 90]])
            assert np.array_equal(e.process_host(batch).view(np.uint32), got[c0:c0 + 90].view(np.uint32)), (rank, c0)
        assert e.lib.wrp_last_hip_error(e.handle) == b""
