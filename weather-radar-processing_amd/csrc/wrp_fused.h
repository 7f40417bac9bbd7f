// wrp_fused.h -- fused persistent launch: both passes in ONE launch, the 2 MiB intermediate of a
// sector-channel stays in an XCD's L2 (ONE buffer per XCD: two of them plus the streaming input
// were measured to thrash the 4 MiB L2 -- every row then came back over the fabric; and the input
// is read with non-temporal loads, or it pushes the buffer out all the same).
//
// Geometry, from the tile width TCOLS (8 or 16 columns):
//   workgroup = 64 TCOLS threads (TCOLS waves); tiles per channel-task = 512 / TCOLS = members at
//   work per team; rows per member and task = TCOLS (one per wave).
//   TCOLS = 16: one 1024-thread workgroup per CU (the form the engine launches).  TCOLS = 8: two
//   512-thread workgroups per CU (80 KiB of LDS each); measured slower (5.9 vs 3.9 us/sector: the
//   range stages are latency-bound per wave) and not wired to a flag -- the engine's
//   8-column fused launch is wrp_fused_roles.h, which shares the device functions below.
// At start every workgroup reads the XCD it runs on (HW_REG_XCC_ID -- placement is READ, never
// assumed), registers with that XCD's team and the grid meets once.  Team e owns sectors e,
// e + teams, ...; a sector is two channel-TASKS q = 0, 1, 2, ...; member r owns range tile r and the
// Doppler rows TCOLS r .. TCOLS r + TCOLS - 1 of every task.  The work is software-pipelined so that
// neither hand-off is waited for right after it is produced.  Round t of a workgroup:
//     S(t)   stages 1-3 of its tile of task t; the 4 output float4 per lane STAY IN REGISTERS
//     B(t-1) its rows of task t-1: needs every tile of t-1 stored      (counter `stored`, counted
//            by the members during their S(t) -- a stage of arithmetic ago)
//     W(t)   store the tile of task t into the team's buffer: needs every row of t-1 loaded
//            (counter `loaded`, counted by each wave when its row has arrived -- a row transform ago)
// Both dependencies point backwards and every workgroup walks the rounds in order, so the
// workgroup that is furthest behind can always run: no cycle.  Latency hiding:
//   * a counter is read by one lane while the other waves still compute, and only if it was not
//     yet satisfied does the workgroup fall into the (bounded) polling loop;
//   * the tile stores drain while the next round's stage 1 computes; their count is added at that
//     stage's barrier -- except that a workgroup never polls with an unsent count (it flushes
//     first), which keeps the no-cycle argument valid;
//   * the next tile is requested right after stage 1 has consumed the current one, a whole round
//     before it is needed.
// All spins are bounded and a timeout is reported.  Cache behaviour, all measured on MI355X
// (tools/l2handoff.hip, FETCH_SIZE / WRITE_SIZE): tiles are stored with plain stores and stay in
// the XCD's L2; rows are read with sc1 loads, which miss the reader's L1 (an sc0 load does not) and
// are served by that L2 at 1.5 TB/s per XCD; team counters are L2 atomics of workgroup scope (no
// sc1), read back with an atomic add of 0.  HH row sums wait in a register for the VV task of the
// same gate (same wave, next round).
//
// Range FFT: 1024 = 8 x 16 x 8, in-place DIF over positions p of a column:
//   stage 1 (registers, from the prefetch): lane owns rows p0 + 128 r, r < 8, of two columns
//           -> radix 8, twiddle W_1024^{p0 k1}, to LDS position k1*128 + p0
//   stage 2 (LDS, one column per lane, b64): positions k1*128 + p1 + 8 r, r < 16 -> radix 16,
//           twiddle W_128^{p1 k2}, in place
//   stage 3 (LDS, column pair per lane, b128): positions k1*128 + k2*8 + r, r < 8 -> radix 8;
//           gate k = k1 + 8 k2 + 128 k3, k3 < 4 kept.
// Same padded LDS image as range_pass_1024<TCOLS> (RangeTile<TCOLS>), twiddles in its padding.
// Arithmetic differs from the two-kernel path only in the factorisation of the range FFT
// (8x16x8 instead of 16x8x8), so results agree to rounding, not bit for bit; the 8- and the
// 16-column geometry perform identical arithmetic per element and agree bit for bit.
#pragma once
#include <hip/hip_runtime.h>

#include "wrp_kernels.h"

namespace wrp {

constexpr int FUSED_RING = 8;   // completion counters are a ring over tasks (at most 2 tasks are in flight)
struct FusedCtl {               // zeroed by hipMemsetAsync before every launch; every counter on its own 64-byte line
    unsigned census[8];         // workgroups per XCC
    unsigned arrived;           // grid-wide start counter
    unsigned timeout;           // 1: a bounded spin gave up; 2: a team has too few workgroups
    unsigned pad[6];
    unsigned stored[8][FUSED_RING][16];     // tiles of task q stored: slot q % RING, target ITEMS * (q / RING + 1)
    unsigned loaded[8][FUSED_RING][16];     // rows of task q in registers: target 512 * (q / RING + 1)
    unsigned census_rows[8];                // fused_roles: row workgroups per XCC (census[] counts the tile workgroups)
    unsigned cu_arrivals[8][256];           // fused64: workgroups seen per physical CU (key = HW_ID bits 15:8: se, sh, cu)
};
constexpr int FUSED_STAMP_TASKS = 16;
constexpr size_t FUSED_MID_ELEMS = (size_t)(RP_M / 2) * DP_N;              // one channel
constexpr size_t FUSED_TEAM_ELEMS = FUSED_MID_ELEMS;                       // per team: ONE mid buffer (float2 units)

template <int TCOLS>
struct FusedGeom {
    typedef RangeTile<TCOLS> FT;
    static constexpr int CP = TCOLS / 2;                  // column pairs
    static constexpr int WAVES = TCOLS;
    static constexpr int THREADS = 64 * WAVES;            // 1024 / 512
    static constexpr int ITEMS = DP_N / TCOLS;            // tiles per task = members at work per team: 32 / 64
    static constexpr int WG_PER_CU = 16 / TCOLS;          // 1 / 2
    // LDS: image (twiddles in its pads) | window wr_c[1024] | Doppler twiddles [512], whose entries
    // 448.. are never read (largest index used: 63 * 7 = 441) and hold the control words
    static constexpr int OFF_TWN = FT::LDS_BYTES;
    static constexpr int OFF_CTL = OFF_TWN + 448 * 8;
    static constexpr int LDS_BYTES = OFF_TWN + DP_N * 8;  // 155648 / 81920 = exactly half of the CU's 160 KiB
    // the row buffers (one per wave) alias the image from block 0; they overwrite the twiddle pads of
    // those blocks unless the twiddles live above them (16 columns), else the pads are re-filled
    static constexpr int ROW_BLOCKS = WAVES * DP_ELEMS * 8 / FT::BLK_BYTES;
    static constexpr bool TW_CLOBBERED = FT::TW_BLK0 < ROW_BLOCKS;
    static constexpr int TW_LOST = TW_CLOBBERED ? ROW_BLOCKS * FT::TW_PER_PAD : 0;   // entries 0 .. TW_LOST-1
    static_assert(WAVES * DP_ELEMS * 8 % FT::BLK_BYTES == 0, "row buffers end on a block boundary");
    static_assert(LDS_BYTES * WG_PER_CU <= 160 * 1024, "fused launch exceeds the CU's LDS");
    static_assert(TW_LOST <= THREADS, "one twiddle per thread is re-filled");
};

__device__ __forceinline__ unsigned xcc_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7;
}

// Team counters live in the XCD's L2 and are only ever touched by that XCD's workgroups, so they
// need no coherence beyond it: additions are plain L2 atomics (workgroup scope, no sc1).
__device__ __forceinline__ void l2_count(unsigned *p)
{
    __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ unsigned l2_peek(unsigned *p)
{
    // An atomic add of 0 with return: performed at the L2 like every atomic, so it cannot hit a stale
    // line of the CU's L1.  The zero is hidden from the compiler, which otherwise folds the idempotent
    // atomic into an sc0 load.  Every alternative was measured worse on MI355X (the rounds launch of
    // this file, us/sector): sc0 load -- never sees the count (stale L1 line); device-scope (sc1) load
    // 5.7; non-temporal load 7.2 (sees it late); this 3.9.  Its weakness: dozens of workgroups polling
    // ONE line serialise in the L2's atomic unit (2.7 us median, 4.8 us worst, to notice a count in the
    // tile/row launches, where 32 workgroups poll for most of a task) -- poll rarely, or per-waiter lines.
    unsigned zero = 0;
    asm volatile("" : "+v"(zero));
    return __hip_atomic_fetch_add(p, zero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// control words in LDS: address space 3 spelled out, because hipcc does not infer it for volatile
// accesses and would emit flat instructions with sc0 sc1 for them
typedef __attribute__((address_space(3))) volatile int lds_word;

// every thread calls; thread 0 polls; false = timed out.  TEAM: the counter is a team counter (see above),
// otherwise it is shared by the whole grid and read at device scope.
template <bool TEAM>
__device__ __forceinline__ bool team_wait_ge(unsigned *p, unsigned target, unsigned *tmo, lds_word *s_ok)
{
    if (threadIdx.x == 0) {
        int good = 0;
#pragma unroll 1
        for (unsigned spins = 0; spins < (1u << 21); spins++) {
            const unsigned now = TEAM ? l2_peek(p) : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (now >= target) { good = 1; break; }
            if ((spins & 63) == 63 && __hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
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
template <int TCOLS, int AUX = StreamAux<TCOLS>::value>
__device__ __forceinline__ void fused_tile_load(const float2 *src /* wave-uniform */, int n, int col_base, const float *wd,
                                                float4 (&v)[8], float2 &wdv, bool valid)
{
    typedef FusedGeom<TCOLS> G;
    const int w = wave_id();
    int l = threadIdx.x & 63;
    asm volatile("" : "+v"(l));   // recompute the lane offsets per call instead of keeping (spilling) them
    const int p0 = w * (64 / G::CP) + l / G::CP, cp = l % G::CP;
    const rsrc_t rs = make_rsrc(src, valid ? (unsigned)RP_M * n * 8u : 0u);
    const int voff = (p0 * n + col_base + cp * 2) * 8;
#pragma unroll
    // non-temporal: the input streams through the L2 once and must not push the team's buffer out of it
    for (int r = 0; r < 8; r++) v[r] = buf_load_f4<AUX>(rs, voff, 128 * r * n * 8);
    wdv = buf_load_f2<0>(make_rsrc(wd, (unsigned)n * 4u), (col_base + cp * 2) * 4, 0);
}

template <int TCOLS, class Hook0, class Hook>
__device__ __forceinline__ void fused_stage12(unsigned char *smem, float4 (&v)[8], float2 wdv, Hook0 before_barrier1,
                                              Hook after_stage1)
{
    typedef FusedGeom<TCOLS> G;
    typedef typename G::FT FT;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));   // every per-lane LDS address below is recomputed per call, not hoisted + spilled
    {   // ---- stage 1: radix 8 over rows p0 + 128 r, two columns per lane
        const int w = tid >> 6, l = tid & 63, cp = l % G::CP;
        const int p0 = w * (64 / G::CP) + l / G::CP;
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
    before_barrier1();
    __syncthreads();
    after_stage1();   // v has been consumed: the next tile may be requested into it
    {   // ---- stage 2: radix 16 over positions k1*128 + p1 + 8 r, ONE column per lane
        const int col = tid % TCOLS, k1 = tid / (TCOLS * 8);
        const int p1 = (tid / TCOLS) & 7;
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

// stage 3 in two halves: the arithmetic (outputs of gates k1 + 8 k2 + 128 k3, k3 < 4, of one column pair) ...
template <int TCOLS>
__device__ __forceinline__ void fused_stage3_compute(const unsigned char *smem, float4 (&o)[4])
{
    typedef FusedGeom<TCOLS> G;
    typedef typename G::FT FT;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int cp = tid % G::CP, k1 = tid / (G::CP * 16), k2 = (tid / G::CP) & 15;
    cf a[8], c[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const float4 u = *reinterpret_cast<const float4 *>(smem + FT::addr(k1 * 128 + k2 * 8 + r, cp));
        a[r] = make_float2(u.x, u.y);
        c[r] = make_float2(u.z, u.w);
    }
    fft8<-1>(a);
    fft8<-1>(c);
#pragma unroll
    for (int k3 = 0; k3 < 4; k3++) o[k3] = make_float4(a[k3].x, a[k3].y, c[k3].x, c[k3].y);
}

// ... and the stores, which may happen much later
template <int TCOLS, int AUX = 0>
__device__ __forceinline__ void fused_stage3_store(float2 *dst /* wave-uniform */, int n, int col_base, const float4 (&o)[4])
{
    typedef FusedGeom<TCOLS> G;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int cp = tid % G::CP, k1 = tid / (G::CP * 16), k2 = (tid / G::CP) & 15;
    const rsrc_t rd = make_rsrc(dst, (unsigned)(RP_M / 2) * n * 8u);
    const int voff = ((k1 + 8 * k2) * n + col_base + cp * 2) * 8;
#pragma unroll
    for (int k3 = 0; k3 < 4; k3++)   // gates < m/2 only; row offset in the VGPR (see buf_store_f4)
        buf_store_f4<AUX>(rd, voff + 128 * k3 * n * 8, 0, o[k3]);
}

template <int TCOLS, int TAPS>
// 4 waves per SIMD in both geometries (two 8-wave workgroups or one 16-wave workgroup per CU): 128 VGPRs
__global__ __launch_bounds__(FusedGeom<TCOLS>::THREADS, 4) void fused_sector_1024x512(
    const float2 *__restrict__ iq,   // [S][C][1024][512]
    float *__restrict__ out,         // [S][512][2]
    float2 *pool,                    // [8][FUSED_TEAM_ELEMS] per team: mid[512][512]
    FusedCtl *ctl, RangeConsts rc, const float2 *__restrict__ tw_n, int n_sectors, int channels, MaTaps taps,
    float k_rr, float k_cal, unsigned long long *stamps /* diagnostics: [grid][FUSED_STAMP_TASKS][8] or nullptr */)
{
    typedef FusedGeom<TCOLS> G;
    typedef typename G::FT FT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_word *s_ctl = (lds_word *)(smem + G::OFF_CTL);
    float2 *s_twn = reinterpret_cast<float2 *>(smem + G::OFF_TWN);
    const int tid = threadIdx.x, w = wave_id(), l = tid & 63;
    const int n = DP_N, gates = RP_M / 2;
    const DumpPtrs nodump{};
    for (int e = tid; e < 448; e += G::THREADS) s_twn[e] = tw_n[e];     // entries 448.. are the control words
    for (int e = tid; e < RP_M; e += G::THREADS) {   // twiddle table into the image's padding, window behind the image
        *reinterpret_cast<float2 *>(smem + FT::tw_addr(e)) = rc.tw[e];
        reinterpret_cast<float *>(smem + FT::OFF_WR)[e] = rc.wr_c[e];
    }

    // ---- team formation -----------------------------------------------------------------
    if (tid == 0) {
        const unsigned x = xcc_id();
        s_ctl[1] = (int)x;
        s_ctl[2] = (int)atomicAdd(&ctl->census[x], 1u);
        __hip_atomic_fetch_add(&ctl->arrived, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!team_wait_ge<false>(&ctl->arrived, gridDim.x, &ctl->timeout, &s_ctl[0])) return;
    if (tid == 0) {
        int teams = 0, trank = 0;
        for (int x = 0; x < 8; x++) {
            const unsigned c = __hip_atomic_load(&ctl->census[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (c) { if (x < s_ctl[1]) trank++; teams++; }
            if (x == s_ctl[1]) s_ctl[7] = (int)c;
        }
        s_ctl[4] = teams;
        s_ctl[5] = trank;
    }
    __syncthreads();
    // wave-uniform by construction; readfirstlane tells the compiler so (scalar address arithmetic)
    const int xcc = __builtin_amdgcn_readfirstlane(s_ctl[1]);
    const int teams = __builtin_amdgcn_readfirstlane(s_ctl[4]), trank = __builtin_amdgcn_readfirstlane(s_ctl[5]);
    const int rank = __builtin_amdgcn_readfirstlane(s_ctl[2]), sz = __builtin_amdgcn_readfirstlane(s_ctl[7]);
    const int T = 2 * ((n_sectors - trank + teams - 1) / teams);             // channel-tasks of this team
    float2 *mid = pool + (size_t)xcc * FUSED_TEAM_ELEMS;
    float2 *wbuf = reinterpret_cast<float2 *>(smem) + (size_t)w * DP_ELEMS;   // aliases the image from block 0 (rows only)
    if (sz < G::ITEMS) {               // the static schedule needs ITEMS members per team
        if (tid == 0) __hip_atomic_store(&ctl->timeout, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (rank >= G::ITEMS) return;      // surplus members own nothing
    const int col0 = rank * TCOLS, gate = rank * G::WAVES + w;

    auto tile_src = [&](int q) { return iq + ((size_t)(trank + (q >> 1) * teams) * channels + (q & 1)) * RP_M * (size_t)n; };
    auto counter = [&](unsigned (*arr)[FUSED_RING][16], int q) { return &arr[xcc][q % FUSED_RING][0]; };
    auto turns = [&](int q) { return (unsigned)(q / FUSED_RING + 1); };
    auto stamp = [&](int round, int k) {
        if (stamps && tid == 0 && round < FUSED_STAMP_TASKS)
            stamps[((size_t)blockIdx.x * FUSED_STAMP_TASKS + round) * 8 + k] = __builtin_amdgcn_s_memrealtime();
    };

    unsigned *pend = nullptr;   // `stored` counter of the tile just written: its stores are still draining
    // stores drained by every wave, then one lane counts; in front of every polling loop
    auto flush = [&]() {
        if (pend) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) l2_count(pend);
            pend = nullptr;
        }
    };
    int chk = 0;
    // dependency `*p >= tgt`, of which lane 0 holds a recent value: one barrier when it was already
    // satisfied, otherwise flush the unsent count and poll
    auto resolve = [&](unsigned seen, unsigned *p, unsigned tgt) -> bool {
        lds_word *slot = s_ctl + 8 + (chk++ & 3);
        if (tid == 0) *slot = seen >= tgt;
        __syncthreads();
        if (*slot) return true;
        flush();
        return team_wait_ge<true>(p, tgt, &ctl->timeout, &s_ctl[0]);
    };

    float4 v[8];         // this lane's share of the tile of the coming round
    float2 wdv;
    float s_hh = 0.f;    // HH row sum of this wave's gate, waiting for the VV task
    fused_tile_load<TCOLS>(tile_src(0), n, col0, rc.wd, v, wdv, T > 0);
#pragma unroll 1
    for (int t = 0; t <= T; t++) {
        float4 o[4];
        unsigned seen = 0;
        stamp(t, 0);
        if (t < T) {
            // ---------------- S(t): this member's tile of task t, outputs kept in registers ----------------
            const float2 wcur = wdv;
            if (stamps) {   // diagnostics only: when did the tile arrive
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                stamp(t, 1);
            }
            fused_stage12<TCOLS>(
                smem, v, wcur,
                [&]() {   // before the barrier after stage 1: the previous tile's stores have drained behind the arithmetic
                    if (pend) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                },
                [&]() {   // after it: count that tile, request the next one
                    if (pend && tid == 0) l2_count(pend);
                    pend = nullptr;
                    fused_tile_load<TCOLS>(tile_src(t + 1 < T ? t + 1 : 0), n, col0, rc.wd, v, wdv, t + 1 < T);
                });
            stamp(t, 2);
            if (t >= 1 && tid == 0) seen = l2_peek(counter(ctl->stored, t - 1));   // looked at after stage 3
            fused_stage3_compute<TCOLS>(smem, o);
        } else {
            flush();   // last round: the last tile's count is still unsent
            if (tid == 0) seen = l2_peek(counter(ctl->stored, t - 1));
        }
        if (t >= 1) {
            // ---------------- B(t-1): this member's rows of task t-1 ----------------
            const int q = t - 1;
            // every tile of the task stored (also the barrier that frees the image for the row buffers)
            if (!resolve(seen, counter(ctl->stored, q), G::ITEMS * turns(q))) return;
            stamp(t, 3);
            cf x[8];
            doppler_load_row<AUX_SC1>(mid + (size_t)gate * n, l, x);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the row is in registers: the buffer may be overwritten
            if (l == 0) l2_count(counter(ctl->loaded, q));
            stamp(t, 4);
            const float s = doppler_row<false, TAPS>(x, wbuf, s_twn, taps, l, gate, false, nodump);
            stamp(t, 5);
            if ((q & 1) == 0) s_hh = s;
            else if (l == 0) reflectivity_store(&out[((size_t)(trank + (q >> 1) * teams) * gates + gate) * 2], gate, s_hh, s, k_rr, k_cal);
        } else {
            __syncthreads();
        }
        if (t < T) {
            // ---------------- W(t): store the tile; every row of task t-1 must have been loaded ----------------
            if (t >= 1) {
                if (tid == 0) seen = l2_peek(counter(ctl->loaded, t - 1));   // wave 0 is done first; the others still compute
                if (!resolve(seen, counter(ctl->loaded, t - 1), DP_N * turns(t - 1))) return;
                if (G::TW_CLOBBERED) {   // the row buffers have overwritten twiddle pads: re-fill them (L1/L2 hits) ...
                    if (tid < G::TW_LOST) *reinterpret_cast<float2 *>(smem + FT::tw_addr(tid)) = rc.tw[tid];
                }
            }
            stamp(t, 6);
            fused_stage3_store<TCOLS>(mid, n, col0, o);
            pend = counter(ctl->stored, t);
            if (G::TW_CLOBBERED) __syncthreads();   // ... before the next stage 1 reads them
        }
        stamp(t, 7);
    }
    flush();
}

} // namespace wrp
