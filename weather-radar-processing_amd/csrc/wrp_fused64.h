// wrp_fused64.h -- fused persistent launch at EIGHT waves per SIMD.
//
// The range stages of wrp_fused.h are latency-bound per wave: a wave owns 8 rows of
// two columns, needs ~105 VGPRs, and only four such waves fit a SIMD.  Here a lane owns 8 rows of
// ONE column, every stage keeps 1024 lanes busy on an 8-column tile, and the whole launch fits 64
// VGPRs: 2 x 16 waves per CU.  Same team protocol as wrp_fused_roles.h, with 1024-thread workgroups:
//   tile workgroups (first half of the grid): member r < 32 of an XCD's team transforms the two
//       8-column tiles 2r, 2r + 1 of every channel-task into the team's ONE L2-resident buffer;
//   row workgroups (second half): member r < 32 transforms the rows 16 r + w (wave w < 16).
// Counters stored[q] / loaded[q] as there (L2 atomics); all spins bounded.
//
// Range FFT 1024 = 8 x 8 x 16, in-place DIF over positions p of a column, one column per lane:
//   stage 1 (registers, from the prefetch): rows p0 + 128 r, r < 8 -> radix 8, twiddle
//           W_1024^{p0 k1}, to position k1*128 + p0;
//   stage 2 (LDS): positions k1*128 + p1 + 16 r, r < 8, p1 < 16 -> radix 8, twiddle W_128^{p1 k2},
//           in place (result at k1*128 + p1 + 16 k2);
//   stage 3 (LDS): the 16 contiguous positions k1*128 + 16 k2 + r -> 16-point DFT of which only the
//           outputs k3 < 8 are wanted (gate k = k1 + 8 k2 + 64 k3 < 512).  Two lanes share a group:
//           waves 0-7 produce the even k3 (8-point DFT of x[r] + x[r+8], outputs 0..3), waves 8-15
//           the odd k3 (8-point DFT of (x[r] - x[r+8]) W_16^r, outputs 0..3).
// LDS image: [position][8 columns] complex = 64 bytes per position, 32 bytes of padding after every 8
// positions: consecutive groups of 16 positions then start 64 bytes apart modulo the 128-byte bank
// window, which makes stage 3's reads (lanes = 8 columns x 8 groups) conflict-free; stages 1 and 2
// touch 512 contiguous bytes per wave-instruction.  The padding holds the range window (8 floats
// per pad); the twiddle table sits behind the image.
// Results agree with the other paths to rounding (yet another factorisation); bit-reproducible.
//
// STATUS: experimental (engine: WRP_FUSED64=1 together with WRP_FLAG_FUSED), correct (tested), 64 VGPRs
// with 2 spills, and measured 5.7 us/sector -- slower than every other path.  The stamps
// (tools/roles_stamps.py) say why, and none of it is the occupancy this file was written to fix:
//   * an 8-column tile covers HALF of every 128-byte line; read non-temporally the other half is gone
//     when the workgroup's second tile asks for it 6 us later, so the input is fetched twice
//     (FETCH_SIZE: 15.9 MiB/sector) and the tile requests of the 32 synchronised tile workgroups of a
//     team are served at the HBM fair share: the waves sit 3.8 us in the ISSUE of the request (the CU's
//     memory pipeline takes a bounded number of outstanding lines), which is accounted to stage 2;
//   * 32 workgroups polling one counter line with atomics (l2_peek) take 2.7 us (median) to notice a
//     count, twice per task;
//   * a tile workgroup can be only one tile ahead of the single buffer, so its two tiles serialise.
// Arithmetic itself: stage 1 1.3 us, stage 3 0.4 us per tile.  The next step is a 16-column variant of
// this factorisation (full lines, one request burst per task) with per-waiter flag lines.
#pragma once
#include "wrp_fused.h"

namespace wrp {

struct F64 {
    static constexpr int THREADS = 1024, WAVES = 16, MEMBERS = 32, TCOLS = 8;
    static constexpr int ROW_BYTES = 64, BLK_BYTES = 8 * ROW_BYTES + 32, IMG_BYTES = (RP_M / 8) * BLK_BYTES;   // 69632
    static constexpr int OFF_TW = IMG_BYTES;                       // float2 [1024] exp(-2 pi i k / 1024)
    static constexpr int OFF_CTL = OFF_TW + RP_M * 8;              // 77824, both kinds
    static constexpr int LDS_BYTES = OFF_CTL + 64;                 // 77888; two workgroups per CU
    static constexpr int OFF_TWN = WAVES * DP_ELEMS * 8;           // row kind: Doppler twiddles behind the 16 row buffers
    static_assert(OFF_TWN + DP_N * 8 <= OFF_CTL, "row workgroup layout fits");
    static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
    static __device__ __forceinline__ int addr(int pos, int col) { return (pos >> 3) * BLK_BYTES + (pos & 7) * ROW_BYTES + col * 8; }
    static __device__ __forceinline__ int wr_addr(int e) { return (e >> 3) * BLK_BYTES + 8 * ROW_BYTES + (e & 7) * 4; }
};

// this lane's 8 row loads (rows p0 + 128 r of one column) + its Doppler-window value
template <int AUX>
__device__ __forceinline__ void f64_tile_load(const float2 *src /* wave-uniform */, int n, int col_base, const float *wd,
                                              cf (&v)[8], float &wdv, bool valid)
{
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int col = tid & 7, p0 = tid >> 3;
    const rsrc_t rs = make_rsrc(src, valid ? (unsigned)RP_M * n * 8u : 0u);
    const int voff = (p0 * n + col_base + col) * 8;
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = buf_load_f2<AUX>(rs, voff, 128 * r * n * 8);
    wdv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(make_rsrc(wd, (unsigned)n * 4u), (col_base + col) * 4, 0, 0));
}

template <class Hook>
__device__ __forceinline__ void f64_stage12(unsigned char *smem, cf (&v)[8], float wdv, Hook after_stage1)
{
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const float2 *s_tw = reinterpret_cast<const float2 *>(smem + F64::OFF_TW);
    {   // ---- stage 1: radix 8 over rows p0 + 128 r of one column
        const int col = tid & 7, p0 = tid >> 3;
        cf a[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const float wgt = *reinterpret_cast<const float *>(smem + F64::wr_addr(p0 + 128 * r)) * wdv;
            a[r] = make_float2(v[r].x * wgt, v[r].y * wgt);
        }
        fft8<-1>(a);
        *reinterpret_cast<float2 *>(smem + F64::addr(p0, col)) = a[0];
#pragma unroll
        for (int k1 = 1; k1 < 8; k1++)
            *reinterpret_cast<float2 *>(smem + F64::addr(k1 * 128 + p0, col)) = cmul(a[k1], s_tw[(p0 * k1) & (RP_M - 1)]);
    }
    __syncthreads();
    after_stage1();   // v has been consumed: the next tile may be requested into it
    {   // ---- stage 2: radix 8 over positions k1*128 + p1 + 16 r
        const int col = tid & 7, p1 = (tid >> 3) & 15, k1 = tid >> 7;
        unsigned char *base = smem + F64::addr(k1 * 128 + p1, col);
        cf x[8];
#pragma unroll
        for (int r = 0; r < 8; r++) x[r] = *reinterpret_cast<const float2 *>(base + 2 * r * F64::BLK_BYTES);   // 16 positions = 2 blocks
        fft8<-1>(x);
        *reinterpret_cast<float2 *>(base) = x[0];
#pragma unroll
        for (int k2 = 1; k2 < 8; k2++)
            *reinterpret_cast<float2 *>(base + 2 * k2 * F64::BLK_BYTES) = cmul(x[k2], s_tw[(8 * p1 * k2) & (RP_M - 1)]);
    }
    __syncthreads();
}

// stage 3: half of a 16-point DFT per lane (see the header); o[j] = gate k1 + 8 k2 + 64 h + 128 j
__device__ __forceinline__ void f64_stage3_compute(const unsigned char *smem, cf (&o)[4])
{
    constexpr float c1 = 0.92387953251128675613f; // cos(pi/8)
    constexpr float s1 = 0.38268343236508977173f; // sin(pi/8)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int col = tid & 7, k2 = (tid >> 3) & 7, k1 = (tid >> 6) & 7;
    const bool odd = wave_id() >= 8;          // wave-uniform: no divergence inside a wave
    const unsigned char *base = smem + F64::addr(k1 * 128 + 16 * k2, col);
    cf t[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const cf lo = *reinterpret_cast<const float2 *>(base + r * F64::ROW_BYTES);
        const cf hi = *reinterpret_cast<const float2 *>(base + F64::BLK_BYTES + r * F64::ROW_BYTES);
        t[r] = odd ? csub(lo, hi) : cadd(lo, hi);
    }
    if (odd) {   // times W_16^r, r = 1..7
        t[1] = cmul(t[1], make_float2(c1, -s1));
        t[2] = mul_w8_1<-1>(t[2]);
        t[3] = cmul(t[3], make_float2(s1, -c1));
        t[4] = mul_si<-1>(t[4]);
        t[5] = cmul(t[5], make_float2(-s1, -c1));
        t[6] = mul_w8_3<-1>(t[6]);
        t[7] = cmul(t[7], make_float2(-c1, -s1));
    }
    fft8<-1>(t);
#pragma unroll
    for (int j = 0; j < 4; j++) o[j] = t[j];
}

__device__ __forceinline__ void f64_stage3_store(float2 *dst /* wave-uniform */, int n, int col_base, const cf (&o)[4])
{
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int col = tid & 7, k2 = (tid >> 3) & 7, k1 = (tid >> 6) & 7, h = tid >> 9;
    const rsrc_t rd = make_rsrc(dst, (unsigned)(RP_M / 2) * n * 8u);
    const int voff = ((k1 + 8 * k2 + 64 * h) * n + col_base + col) * 8;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        v2f t;
        t.x = o[j].x; t.y = o[j].y;
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, t), rd, voff + 128 * j * n * 8, 0, 0);
    }
}

template <int TAPS>
__global__ __launch_bounds__(F64::THREADS, 8) void fused64_1024x512(
    const float2 *__restrict__ iq,   // [S][C][1024][512]
    float *__restrict__ out,         // [S][512][2]
    float2 *pool,                    // [8][FUSED_TEAM_ELEMS] per team: mid[512][512]
    FusedCtl *ctl, RangeConsts rc, const float2 *__restrict__ tw_n, int n_sectors, int channels, MaTaps taps,
    float k_rr, float k_cal, unsigned long long *stamps /* diagnostics: [grid][FUSED_STAMP_TASKS][8] or nullptr */)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_word *s_ctl = (lds_word *)(smem + F64::OFF_CTL);
    const int tid = threadIdx.x, w = wave_id(), l = tid & 63;
    const int n = DP_N, gates = RP_M / 2;

    // ---- kind and team: the FIRST workgroup to arrive on a physical CU becomes its tile workgroup, the
    // second its row workgroup (two fit a CU), so that every CU runs one of each -- with the kind taken
    // from blockIdx some CUs got two tile workgroups and set the pace for their whole team (measured).
    // Census per kind and XCD, one grid-wide meeting.
    if (tid == 0) {
        const unsigned x = xcc_id();
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        const unsigned slot = atomicAdd(&ctl->cu_arrivals[x][(hw >> 8) & 255], 1u);
        const bool rows = (slot & 1) != 0;
        s_ctl[1] = (int)x;
        s_ctl[3] = rows ? 1 : 0;
        s_ctl[2] = (int)atomicAdd(rows ? &ctl->census_rows[x] : &ctl->census[x], 1u);
        __hip_atomic_fetch_add(&ctl->arrived, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const bool rows_kind = __builtin_amdgcn_readfirstlane(s_ctl[3]) != 0;   // wave-uniform
    if (!team_wait_ge<false>(&ctl->arrived, gridDim.x, &ctl->timeout, &s_ctl[0])) return;
    if (tid == 0) {
        int teams = 0, trank = 0, ok = 1;
        for (int x = 0; x < 8; x++) {
            const unsigned ct = __hip_atomic_load(&ctl->census[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned cr = __hip_atomic_load(&ctl->census_rows[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ct || cr) {
                if (x < s_ctl[1]) trank++;
                teams++;
                if (ct < (unsigned)F64::MEMBERS || cr < (unsigned)F64::MEMBERS) ok = 0;   // a team needs 32 of each kind
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
    if (rank >= F64::MEMBERS) return;   // surplus members own nothing

    auto counter = [&](unsigned (*arr)[FUSED_RING][16], int q) { return &arr[xcc][q % FUSED_RING][0]; };
    auto turns = [&](int q) { return (unsigned)(q / FUSED_RING + 1); };
    auto stamp = [&](int slot, int k) {
        if (stamps && tid == 0 && slot < FUSED_STAMP_TASKS)
            stamps[((size_t)blockIdx.x * FUSED_STAMP_TASKS + slot) * 8 + k] = __builtin_amdgcn_s_memrealtime();
    };

    if (stamps && tid == 0) stamps[(size_t)blockIdx.x * FUSED_STAMP_TASKS * 8 + 5] = rows_kind ? 2 : 1;   // kind, for the tools

    if (!rows_kind) {
        // =============== tile workgroup: tiles 2 rank, 2 rank + 1 of every task =========================
        *reinterpret_cast<float2 *>(smem + F64::OFF_TW + tid * 8) = rc.tw[tid];
        *reinterpret_cast<float *>(smem + F64::wr_addr(tid)) = rc.wr_c[tid];
        auto tile_src = [&](int q) { return iq + ((size_t)(trank + (q >> 1) * teams) * channels + (q & 1)) * RP_M * (size_t)n; };
        cf v[8];
        float wdv;
        f64_tile_load<AUX_NT>(tile_src(0), n, 2 * rank * 8, rc.wd, v, wdv, T > 0);
        __syncthreads();
#pragma unroll 1
        for (int i = 0; i < 2 * T; i++) {   // item i = tile (i & 1) of task i >> 1
            const int q = i >> 1, h = i & 1, col0 = (2 * rank + h) * 8;
            const int nq = (i + 1) >> 1, nh = (i + 1) & 1;
            stamp(i, 0);
            const float wcur = wdv;
            if (stamps) {   // diagnostics only: when did the tile arrive
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                stamp(i, 4);
            }
            f64_stage12(smem, v, wcur, [&]() {
                stamp(i, 6);
                f64_tile_load<AUX_NT>(tile_src(nq < T ? nq : 0), n, (2 * rank + nh) * 8, rc.wd, v, wdv, nq < T);
            });
            stamp(i, 7);
            cf o[4];
            f64_stage3_compute(smem, o);
            stamp(i, 1);
            // the buffer still holds task q-1 until every row of it has been loaded (checked once per task;
            // the barrier inside also separates stage 3's LDS reads from the next stage 1's writes)
            if (h == 0 && q >= 1) {
                if (!team_wait_ge<true>(counter(ctl->loaded, q - 1), DP_N * turns(q - 1), &ctl->timeout, &s_ctl[0])) return;
            } else {
                __syncthreads();
            }
            stamp(i, 2);
            f64_stage3_store(mid, n, col0, o);
            if (h == 1) {   // both tiles of the task: wait until they are in the L2, then count
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) l2_count(counter(ctl->stored, q));
            }
            stamp(i, 3);
        }
    } else {
        // =============== row workgroup: row 16 rank + w of every task ====================================
        float2 *s_twn = reinterpret_cast<float2 *>(smem + F64::OFF_TWN);
        if (tid < DP_N) s_twn[tid] = tw_n[tid];
        float2 *wbuf = reinterpret_cast<float2 *>(smem) + (size_t)w * DP_ELEMS;
        const DumpPtrs nodump{};
        const int gate = rank * F64::WAVES + w;
        float s_hh = 0.f;   // HH row sum waiting for the VV task
        __syncthreads();
#pragma unroll 1
        for (int q = 0; q < T; q++) {
            stamp(q, 0);
            if (!team_wait_ge<true>(counter(ctl->stored, q), F64::MEMBERS * turns(q), &ctl->timeout, &s_ctl[0])) return;
            stamp(q, 1);
            cf x[8];
            doppler_load_row<AUX_SC1>(mid + (size_t)gate * n, l, x);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the row is in registers: the buffer may be overwritten
            if (l == 0) l2_count(counter(ctl->loaded, q));
            stamp(q, 2);
            const float s = doppler_row<false, TAPS>(x, wbuf, s_twn, taps, l, gate, false, nodump);
            stamp(q, 3);
            if ((q & 1) == 0) s_hh = s;
            else if (l == 0) reflectivity_store(&out[((size_t)(trank + (q >> 1) * teams) * gates + gate) * 2], gate, s_hh, s, k_rr, k_cal);
        }
    }
}

} // namespace wrp
