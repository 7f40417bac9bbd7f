// wrp_fused_roles.h -- fused persistent launch with TWO KINDS of workgroup, so that the LDS- and
// barrier-heavy range stages and the VALU-heavy Doppler rows run CONCURRENTLY on every CU and fill
// each other's stalls (in wrp_fused.h's launch one workgroup does them one after the other and
// leaves the VALU about half used).
//
// Grid = 2 workgroups of 512 threads per CU (78 KiB of LDS each, 4 waves per SIMD in total):
//   tile workgroups (first half of the grid): member r < 32 of an XCD's team transforms the two
//       8-column range tiles 2r and 2r + 1 of every channel-task (same 8x16x8 factorisation and
//       device functions as wrp_fused.h) and stores them into the team's ONE 2 MiB buffer, which
//       stays in that XCD's L2;
//   row workgroups (second half): member r < 32 transforms the Doppler rows 16 r + w and
//       16 r + 8 + w (wave w < 8) of every task from that buffer and writes Zdb / Zdr (the HH row
//       sums wait in registers for the VV task).
// Hand-offs through two team counters (L2 atomics, see wrp_fused.h):
//   stored[q]: a tile workgroup counts once both its tiles of task q are in the L2 (stores drained);
//              the row workgroups load the rows of task q when all 32 have counted;
//   loaded[q]: every wave counts when its two rows are in registers; the tile workgroups store
//              their tiles of task q + 1 when all 256 have counted (the buffer is free).
// The tile side computes a tile of task q + 1 (5 us) while the rows of task q are transformed, and
// only its STORES wait for loaded[q], which completes a microsecond after stored[q]: in steady state
// nobody waits.  stored[0] has no dependency and every later count depends on an earlier one: no
// cycle.  Placement is read (HW_REG_XCC_ID), never assumed; which CU a workgroup lands on only
// matters for speed.  All spins are bounded, a timeout or an undersized team is reported.
// Results are bit-identical to wrp_fused.h's launch (same arithmetic per element).
#pragma once
#include "wrp_fused.h"

namespace wrp {

struct FusedRoles {
    typedef RangeTile<8> FT;
    static constexpr int THREADS = 512, WAVES = 8, MEMBERS = 32;
    static constexpr int OFF_CTL = FT::LDS_BYTES;                 // both kinds: control words behind the tile image
    static constexpr int LDS_BYTES = OFF_CTL + 64;                // 77888; two per CU
    static constexpr int OFF_TWN = WAVES * DP_ELEMS * 8;          // row kind: Doppler twiddles behind the 8 row buffers
    static_assert(OFF_TWN + DP_N * 8 <= OFF_CTL, "row workgroup layout fits");
    static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
};

template <int TAPS, int AUX_IN>
__global__ __launch_bounds__(FusedRoles::THREADS, 4) void fused_roles_1024x512(
    const float2 *__restrict__ iq,   // [S][C][1024][512]
    float *__restrict__ out,         // [S][512][2]
    float2 *pool,                    // [8][FUSED_TEAM_ELEMS] per team: mid[512][512]
    FusedCtl *ctl, RangeConsts rc, const float2 *__restrict__ tw_n, int n_sectors, int channels, MaTaps taps,
    float k_rr, float k_cal, unsigned long long *stamps /* diagnostics: [grid][FUSED_STAMP_TASKS][8] or nullptr */)
{
    typedef FusedRoles R;
    typedef R::FT FT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_word *s_ctl = (lds_word *)(smem + R::OFF_CTL);
    const int tid = threadIdx.x, w = wave_id(), l = tid & 63;
    const int n = DP_N, gates = RP_M / 2;
    const bool rows_kind = blockIdx.x >= gridDim.x / 2;   // wave-uniform, from an SGPR

    // ---- team formation (both kinds): census per kind and XCD, one grid-wide meeting -------------
    if (tid == 0) {
        const unsigned x = xcc_id();
        s_ctl[1] = (int)x;
        s_ctl[2] = (int)atomicAdd(rows_kind ? &ctl->census_rows[x] : &ctl->census[x], 1u);
        __hip_atomic_fetch_add(&ctl->arrived, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!team_wait_ge<false>(&ctl->arrived, gridDim.x, &ctl->timeout, &s_ctl[0])) return;
    if (tid == 0) {
        int teams = 0, trank = 0, ok = 1;
        for (int x = 0; x < 8; x++) {
            const unsigned ct = __hip_atomic_load(&ctl->census[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned cr = __hip_atomic_load(&ctl->census_rows[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ct || cr) {
                if (x < s_ctl[1]) trank++;
                teams++;
                if (ct < (unsigned)R::MEMBERS || cr < (unsigned)R::MEMBERS) ok = 0;   // a team needs 32 of each kind
            }
        }
        s_ctl[4] = teams;
        s_ctl[5] = trank;
        s_ctl[7] = ok;
        if (!ok) __hip_atomic_store(&ctl->timeout, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_ctl[7]) return;
    const int xcc = __builtin_amdgcn_readfirstlane(s_ctl[1]), rank = __builtin_amdgcn_readfirstlane(s_ctl[2]);
    const int teams = __builtin_amdgcn_readfirstlane(s_ctl[4]), trank = __builtin_amdgcn_readfirstlane(s_ctl[5]);
    const int T = 2 * ((n_sectors - trank + teams - 1) / teams);             // channel-tasks of this team
    float2 *mid = pool + (size_t)xcc * FUSED_TEAM_ELEMS;
    if (rank >= R::MEMBERS) return;   // surplus members own nothing

    auto counter = [&](unsigned (*arr)[FUSED_RING][16], int q) { return &arr[xcc][q % FUSED_RING][0]; };
    auto turns = [&](int q) { return (unsigned)(q / FUSED_RING + 1); };
    auto stamp = [&](int slot, int k) {
        if (stamps && tid == 0 && slot < FUSED_STAMP_TASKS)
            stamps[((size_t)blockIdx.x * FUSED_STAMP_TASKS + slot) * 8 + k] = __builtin_amdgcn_s_memrealtime();
    };

    if (!rows_kind) {
        // =============== tile workgroup: tiles 2 rank, 2 rank + 1 of every task =========================
        for (int e = tid; e < RP_M; e += R::THREADS) {   // twiddle table into the image's padding, window behind the image
            *reinterpret_cast<float2 *>(smem + FT::tw_addr(e)) = rc.tw[e];
            reinterpret_cast<float *>(smem + FT::OFF_WR)[e] = rc.wr_c[e];
        }
        auto tile_src = [&](int q) { return iq + ((size_t)(trank + (q >> 1) * teams) * channels + (q & 1)) * RP_M * (size_t)n; };
        float4 v[8];
        float2 wdv;
        fused_tile_load<8, AUX_IN>(tile_src(0), n, 2 * rank * 8, rc.wd, v, wdv, T > 0);
        __syncthreads();
#pragma unroll 1
        for (int i = 0; i < 2 * T; i++) {   // item i = tile (i & 1) of task i >> 1
            const int q = i >> 1, h = i & 1, col0 = (2 * rank + h) * 8;
            const int nq = (i + 1) >> 1, nh = (i + 1) & 1;
            stamp(i, 0);
            const float2 wcur = wdv;
            fused_stage12<8>(smem, v, wcur, [] {},
                             [&]() { fused_tile_load<8, AUX_IN>(tile_src(nq < T ? nq : 0), n, (2 * rank + nh) * 8, rc.wd, v, wdv, nq < T); });
            float4 o[4];
            fused_stage3_compute<8>(smem, o);
            stamp(i, 1);
            // the buffer still holds task q-1 until every row of it has been loaded (checked once per task;
            // the barrier inside also separates stage 3's LDS reads from the next stage 1's writes)
            if (h == 0 && q >= 1) {
                if (!team_wait_ge<true>(counter(ctl->loaded, q - 1), R::MEMBERS * R::WAVES * turns(q - 1), &ctl->timeout, &s_ctl[0])) return;
            } else {
                __syncthreads();
            }
            stamp(i, 2);
            fused_stage3_store<8>(mid, n, col0, o);
            if (h == 1) {   // both tiles of the task: wait until they are in the L2, then count
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) l2_count(counter(ctl->stored, q));
            }
            stamp(i, 3);
        }
    } else {
        // =============== row workgroup: rows 16 rank + w and 16 rank + 8 + w of every task ==============
        float2 *s_twn = reinterpret_cast<float2 *>(smem + R::OFF_TWN);
        for (int e = tid; e < DP_N; e += R::THREADS) s_twn[e] = tw_n[e];
        float2 *wbuf = reinterpret_cast<float2 *>(smem) + (size_t)w * DP_ELEMS;
        const DumpPtrs nodump{};
        const int g0 = rank * 16 + w, g1 = g0 + 8;
        float hh0 = 0.f, hh1 = 0.f;   // HH row sums waiting for the VV task
        __syncthreads();
#pragma unroll 1
        for (int q = 0; q < T; q++) {
            stamp(q, 0);
            if (!team_wait_ge<true>(counter(ctl->stored, q), R::MEMBERS * turns(q), &ctl->timeout, &s_ctl[0])) return;
            stamp(q, 1);
            cf x0[8], x1[8];
            doppler_load_row<AUX_SC1>(mid + (size_t)g0 * n, l, x0);
            doppler_load_row<AUX_SC1>(mid + (size_t)g1 * n, l, x1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // both rows are in registers: the buffer may be overwritten
            if (l == 0) l2_count(counter(ctl->loaded, q));
            stamp(q, 2);
            const float s0 = doppler_row<false, TAPS>(x0, wbuf, s_twn, taps, l, g0, false, nodump);
            const float s1 = doppler_row<false, TAPS>(x1, wbuf, s_twn, taps, l, g1, false, nodump);
            stamp(q, 3);
            if ((q & 1) == 0) {
                hh0 = s0;
                hh1 = s1;
            } else if (l == 0) {
                float *o = out + (size_t)(trank + (q >> 1) * teams) * gates * 2;
                reflectivity_store(o + g0 * 2, g0, hh0, s0, k_rr, k_cal);
                reflectivity_store(o + g1 * 2, g1, hh1, s1, k_rr, k_cal);
            }
        }
    }
}

} // namespace wrp
