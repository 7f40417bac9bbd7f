// wrp_fused.h -- fused persistent launch: one team per XCD, the 2 MiB intermediate of a
// sector-channel stays in that XCD's L2.  1024-thread workgroups (16 waves = 4 per SIMD).
//
// Grid = one workgroup per CU.  At start every workgroup registers with the team of the XCD it
// runs on (HW_REG_XCC_ID -- placement is READ, never assumed) and the grid meets once.  Team e
// then owns sectors e, e + teams, ...; for each of a sector's two channels:
//   A  every member transforms its range tiles (rank, rank + size, ...) and stores them with
//      plain stores into the team's own mid buffer -> the lines stay in this XCD's L2;
//   -- team barrier 1 (device-scope counter; stores drained by every wave first)
//   B  every wave transforms one gate's row, loading it with sc1 loads (bypass the CU's L1, which
//      may hold the previous task's lines of the same addresses; served by the shared L2); right
//      behind its own row loads it requests its share of the NEXT task's tile (vmcnt retires in
//      issue order, so the row is not delayed), which lands during the Doppler arithmetic;
//      HH row sums are parked in LDS, the VV pass finishes Zdb/Zdr;
//   -- team barrier 2 is only waited for just before the NEXT task's stage-3 stores.
// All spins are bounded; a timeout sets ctl->timeout and every workgroup leaves.
//
// Range FFT for 16 waves: 1024 = 8 x 16 x 8, in-place DIF over positions p of a column:
//   stage 1 (registers, from the prefetch): lane owns rows p0 + 128 r, r < 8, of two columns
//           -> radix 8, twiddle W_1024^{p0 k1}, to LDS position k1*128 + p0
//   stage 2 (LDS, one column per lane, b64): positions k1*128 + p1 + 8 r, r < 16 -> radix 16,
//           twiddle W_128^{p1 k2}, in place
//   stage 3 (LDS, column pair per lane, b128): positions k1*128 + k2*8 + r, r < 8 -> radix 8;
//           gate k = k1 + 8 k2 + 128 k3, k3 < 4 stored.
// Same padded LDS image as range_pass_1024<16> (RangeTile<16>), twiddles in its padding.
// Arithmetic differs from the two-kernel path only in the factorisation of the range FFT
// (8x16x8 instead of 16x8x8), so results agree to rounding, not bit for bit.
#pragma once
#include <hip/hip_runtime.h>

#include "wrp_kernels.h"

namespace wrp {

struct FusedCtl {            // zeroed by hipMemsetAsync before every launch
    unsigned census[8];      // workgroups per XCC
    unsigned arrived;        // grid-wide start counter
    unsigned timeout;        // != 0: a bounded spin gave up
    unsigned pad[6];
    unsigned bar1[8][16];    // one 64-byte line per team
    unsigned bar2[8][16];
};
typedef RangeTile<16> FT;
constexpr int FUSED_THREADS = 1024;
constexpr int FUSED_WAVES = 16;
constexpr int FUSED_STAMP_TASKS = 16;
constexpr int FUSED_HH_SLOTS = 32;                                        // gates per wave, worst case (team of one)
constexpr int FUSED_OFF_HH = FT::LDS_BYTES;                                // float [16][FUSED_HH_SLOTS]
constexpr int FUSED_OFF_CTL = FUSED_OFF_HH + FUSED_WAVES * FUSED_HH_SLOTS * 4;   // int [16]
constexpr int FUSED_OFF_TWN = FUSED_OFF_CTL + 64;                          // float2 [512] exp(+2 pi i k / 512)
constexpr int FUSED_LDS_BYTES = FUSED_OFF_TWN + DP_N * 8;                  // 157760 <= 160 KiB
static_assert(FUSED_LDS_BYTES <= 160 * 1024, "fused launch exceeds the CU's LDS");
static_assert(FUSED_WAVES * DP_ELEMS * 8 <= FT::TW_BLK0 * FT::BLK_BYTES, "phase-B wave buffers must stay below the twiddle pads");
constexpr size_t FUSED_MID_ELEMS = (size_t)(RP_M / 2) * DP_N;              // per team, one channel

__device__ __forceinline__ unsigned xcc_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7;
}

// every thread calls; thread 0 polls (relaxed, device scope); false = timed out
__device__ __forceinline__ bool team_wait_ge(unsigned *p, unsigned target, unsigned *tmo, volatile int *s_ok)
{
    if (threadIdx.x == 0) {
        int good = 0;
#pragma unroll 1
        for (unsigned spins = 0; spins < (1u << 21); spins++) {
            if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) { good = 1; break; }
            if (__hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
            __builtin_amdgcn_s_sleep(2);
        }
        if (!good) __hip_atomic_store(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_ok = good;
    }
    __syncthreads();
    const bool ok = *s_ok != 0;
    __syncthreads();
    return ok;
}

// this lane's 8 row loads (rows p0 + 128 r of one column pair) + its two Doppler-window values;
// valid = false -> zero-record descriptor, all loads dropped (see range_load)
__device__ __forceinline__ void fused_tile_load(const float2 *src /* wave-uniform */, int n, int col_base, const float *wd,
                                                float4 (&v)[8], float2 &wdv, bool valid)
{
    const int w = wave_id();
    int l = threadIdx.x & 63;
    asm volatile("" : "+v"(l));   // recompute the lane offsets per call instead of keeping (spilling) them
    const int p0 = w * 8 + (l >> 3);
    const rsrc_t rs = make_rsrc(src, valid ? (unsigned)RP_M * n * 8u : 0u);
    const int voff = (p0 * n + col_base + (l & 7) * 2) * 8;
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = buf_load_f4(rs, voff, 128 * r * n * 8);
    wdv = buf_load_f2<0>(make_rsrc(wd, (unsigned)n * 4u), (col_base + (l & 7) * 2) * 4, 0);
}

__device__ __forceinline__ void fused_stage12(unsigned char *smem, float4 (&v)[8], float2 wdv)
{
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));   // every per-lane LDS address below is recomputed per call, not hoisted + spilled
    {   // ---- stage 1: radix 8 over rows p0 + 128 r, two columns per lane
        const int w = tid >> 6, l = tid & 63, cp = l & 7;
        const int p0 = w * 8 + (l >> 3);
        const float *s_wr = reinterpret_cast<const float *>(smem + FT::OFF_WR);
        cf a[8], c[8], t1[8];
#pragma unroll
        for (int k1 = 1; k1 < 8; k1++) t1[k1] = *reinterpret_cast<const float2 *>(smem + FT::tw_addr((p0 * k1) & (RP_M - 1)));
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const float wrow = s_wr[p0 + 128 * r];
            const float w0 = wrow * wdv.x, w1 = wrow * wdv.y;
            a[r] = make_float2(v[r].x * w0, v[r].y * w0);
            c[r] = make_float2(v[r].z * w1, v[r].w * w1);
        }
        fft8<-1>(a);
        fft8<-1>(c);
        *reinterpret_cast<float4 *>(smem + FT::addr(p0, cp)) = make_float4(a[0].x, a[0].y, c[0].x, c[0].y);
#pragma unroll
        for (int k1 = 1; k1 < 8; k1++) {
            const cf x = cmul(a[k1], t1[k1]), y = cmul(c[k1], t1[k1]);
            *reinterpret_cast<float4 *>(smem + FT::addr(k1 * 128 + p0, cp)) = make_float4(x.x, x.y, y.x, y.y);
        }
    }
    __syncthreads();
    {   // ---- stage 2: radix 16 over positions k1*128 + p1 + 8 r, ONE column per lane
        const int col = tid & 15, k1 = tid >> 7;
        const int p1 = (tid >> 4) & 7;
        unsigned char *base = smem + (col >> 1) * 16 + (col & 1) * 8;
        cf x[16], t2[16];
#pragma unroll
        for (int k2 = 1; k2 < 16; k2++) t2[k2] = *reinterpret_cast<const float2 *>(smem + FT::tw_addr((8 * p1 * k2) & (RP_M - 1)));
#pragma unroll
        for (int r = 0; r < 16; r++) x[r] = *reinterpret_cast<const float2 *>(base + FT::addr(k1 * 128 + p1 + 8 * r, 0));
        fft16<-1>(x);
        *reinterpret_cast<float2 *>(base + FT::addr(k1 * 128 + p1, 0)) = x[0];
#pragma unroll
        for (int k2 = 1; k2 < 16; k2++)
            *reinterpret_cast<float2 *>(base + FT::addr(k1 * 128 + p1 + 8 * k2, 0)) = cmul(x[k2], t2[k2]);
    }
    __syncthreads();
}

__device__ __forceinline__ void fused_stage3(const unsigned char *smem, float2 *dst /* wave-uniform */, int n, int col_base)
{
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int cp = tid & 7, k1 = tid >> 7, k2 = (tid >> 3) & 15;
    cf a[8], c[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const float4 u = *reinterpret_cast<const float4 *>(smem + FT::addr(k1 * 128 + k2 * 8 + r, cp));
        a[r] = make_float2(u.x, u.y);
        c[r] = make_float2(u.z, u.w);
    }
    fft8<-1>(a);
    fft8<-1>(c);
    const rsrc_t rd = make_rsrc(dst, (unsigned)(RP_M / 2) * n * 8u);
    const int voff = ((k1 + 8 * k2) * n + col_base + cp * 2) * 8;
#pragma unroll
    for (int k3 = 0; k3 < 4; k3++)   // gates < m/2 only; row offset in the VGPR (see buf_store_f4)
        buf_store_f4(rd, voff + 128 * k3 * n * 8, 0, make_float4(a[k3].x, a[k3].y, c[k3].x, c[k3].y));
}

template <int TAPS>
__global__ __launch_bounds__(FUSED_THREADS) void fused_sector_1024x512(
    const float2 *__restrict__ iq,   // [S][C][1024][512]
    float *__restrict__ out,         // [S][512][2]
    float2 *mid_pool,                // [8][512][512] one channel-sized buffer per team
    FusedCtl *ctl, RangeConsts rc, const float2 *__restrict__ tw_n, int n_sectors, int channels, MaTaps taps,
    float k_rr, float k_cal, unsigned long long *stamps /* diagnostics: [grid][FUSED_STAMP_TASKS][8] or nullptr */)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // diagnostic phase stamps (100 MHz s_memrealtime); never read by the kernel itself
#define WRP_STAMP(k)                                                                          \
    do {                                                                                      \
        if (stamps && threadIdx.x == 0 && q < FUSED_STAMP_TASKS)                              \
            stamps[((size_t)blockIdx.x * FUSED_STAMP_TASKS + q) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
    float *s_hh = reinterpret_cast<float *>(smem + FUSED_OFF_HH);
    volatile int *s_ctl = reinterpret_cast<volatile int *>(smem + FUSED_OFF_CTL);
    float2 *s_twn = reinterpret_cast<float2 *>(smem + FUSED_OFF_TWN);
    const int tid = threadIdx.x, w = wave_id(), l = tid & 63;
    const int n = DP_N, gates = RP_M / 2, tiles = DP_N / 16;
    const DumpPtrs nodump{};
    if (tid < DP_N) s_twn[tid] = tw_n[tid];
    {   // twiddle table into the image's padding, window behind the image (1024 threads, 1024 entries)
        *reinterpret_cast<float2 *>(smem + FT::tw_addr(tid)) = rc.tw[tid];
        reinterpret_cast<float *>(smem + FT::OFF_WR)[tid] = rc.wr_c[tid];
    }

    // ---- team formation -----------------------------------------------------------------
    if (tid == 0) {
        const unsigned x = xcc_id();
        s_ctl[1] = (int)x;
        s_ctl[2] = (int)atomicAdd(&ctl->census[x], 1u);
        __hip_atomic_fetch_add(&ctl->arrived, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!team_wait_ge(&ctl->arrived, gridDim.x, &ctl->timeout, &s_ctl[0])) return;
    if (tid == 0) {
        int teams = 0, trank = 0;
        for (int x = 0; x < 8; x++) {
            const unsigned c = __hip_atomic_load(&ctl->census[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (c) { if (x < s_ctl[1]) trank++; teams++; }
            if (x == s_ctl[1]) s_ctl[3] = (int)c;
        }
        s_ctl[4] = teams;
        s_ctl[5] = trank;
    }
    __syncthreads();
    // wave-uniform by construction; readfirstlane tells the compiler so (scalar address arithmetic)
    const int xcc = __builtin_amdgcn_readfirstlane(s_ctl[1]), rank = __builtin_amdgcn_readfirstlane(s_ctl[2]);
    const int size = __builtin_amdgcn_readfirstlane(s_ctl[3]), teams = __builtin_amdgcn_readfirstlane(s_ctl[4]);
    const int trank = __builtin_amdgcn_readfirstlane(s_ctl[5]);
    float2 *mid = mid_pool + (size_t)xcc * FUSED_MID_ELEMS;
    unsigned *bar1 = &ctl->bar1[xcc][0], *bar2 = &ctl->bar2[xcc][0];
    float2 *wbuf = reinterpret_cast<float2 *>(smem) + (size_t)w * DP_ELEMS;   // aliases image blocks 0..63 (phase B only)

    float4 v[8];         // this lane's share of one range tile; refilled during phase B for the next task
    float2 wdv;
    bool have = trank < n_sectors && rank < tiles;
    fused_tile_load(iq + (size_t)(have ? trank : 0) * channels * RP_M * (size_t)n, n, rank * 16, rc.wd, v, wdv, have);
    unsigned q = 0;   // channel-tasks this team has completed
#pragma unroll 1
    for (int sec = trank; sec < n_sectors; sec += teams) {
#pragma unroll 1
        for (int ch = 0; ch < 2; ch++, q++) {
            const float2 *src = iq + ((size_t)sec * channels + ch) * RP_M * (size_t)n;
            // ---- A: range tiles of this member -> team mid buffer ----------------------
#pragma unroll 1
            for (int t = rank; t < tiles; t += size) {
                if (!(have && t == rank)) fused_tile_load(src, n, t * 16, rc.wd, v, wdv, true);
                have = false;
                WRP_STAMP(0);
                if (stamps && threadIdx.x == 0 && q < FUSED_STAMP_TASKS)   // shader clock, for MHz = d[6] / d[0]
                    stamps[((size_t)blockIdx.x * FUSED_STAMP_TASKS + q) * 8 + 6] = __builtin_amdgcn_s_memtime();
                fused_stage12(smem, v, wdv);
                WRP_STAMP(1);
                // the previous task's rows must all have been read before they are overwritten
                if (t == rank && q > 0 && !team_wait_ge(bar2, q * (unsigned)size, &ctl->timeout, &s_ctl[0])) return;
                WRP_STAMP(2);
                fused_stage3(smem, mid, n, t * 16);
                __syncthreads();   // LDS image free for the next tile / phase B
            }
            // every storing wave drains its stores, then one lane signals
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            WRP_STAMP(3);
            if (tid == 0) __hip_atomic_fetch_add(bar1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!team_wait_ge(bar1, (q + 1) * (unsigned)size, &ctl->timeout, &s_ctl[0])) return;
            WRP_STAMP(4);
            // ---- B: Doppler rows, one per wave per round ---------------------------------
            const int g0 = rank * FUSED_WAVES + w, gstep = size * FUSED_WAVES;
            const int nsec = ch == 0 ? sec : sec + teams, nch = ch ^ 1;      // next channel-task
            // No load below sits in a conditional block (see range_load).
            auto row = [&](auto prefetch, int g, int slot) {
                cf x[8];
                doppler_load_row<true>(mid + (size_t)g * n, l, x);
                if constexpr (decltype(prefetch)::value) {
                    const bool nv = nsec < n_sectors && rank < tiles;
                    fused_tile_load(iq + ((size_t)(nv ? nsec : sec) * channels + nch) * RP_M * (size_t)n, n, rank * 16,
                                    rc.wd, v, wdv, nv);
                    have = nv;
                }
                const float s = doppler_row<false, TAPS>(x, wbuf, s_twn, taps, l, g, false, nodump);
                if (l == 0) {
                    if (ch == 0) s_hh[w * FUSED_HH_SLOTS + slot] = s;
                    else reflectivity_store(&out[((size_t)sec * gates + g) * 2], g, s_hh[w * FUSED_HH_SLOTS + slot], s, k_rr, k_cal);
                }
            };
            row(TagTrue{}, g0, 0);
#pragma unroll 1
            for (int g = g0 + gstep, slot = 1; g < gates; g += gstep, slot++) row(TagFalse{}, g, slot);
            // all of this workgroup's row loads have completed (their data was consumed)
            __syncthreads();
            WRP_STAMP(5);
            if (tid == 0) __hip_atomic_fetch_add(bar2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
#undef WRP_STAMP
}

} // namespace wrp
