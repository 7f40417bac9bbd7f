// wrp_fused.h -- fused persistent launch: both passes in ONE launch, the 2 MiB intermediate of a
// sector-channel stays in an XCD's L2.  Dataflow form: no team-wide lockstep.
//
// Grid = one 1024-thread workgroup (16 waves = 4 per SIMD) per CU.  At start every workgroup reads
// the XCD it runs on (HW_REG_XCC_ID -- placement is READ, never assumed), registers with that
// XCD's team and the grid meets once.  Team e owns sectors e, e + teams, ...; a sector is two
// channel-TASKS q = 0, 1, 2, ... and a task is 32 A-items (range tiles of 16 columns) followed by
// 32 B-items (16 Doppler rows each, one per wave).  The team's work is the list of item GROUPS
//        A0  A1  B0  A2  B1  A3  B2 ...  A(T-1)  B(T-2)  B(T-1)
// and member r of a team of sz workgroups owns items r, r + sz, ... of every group (a static
// schedule: the next item is known at once, so it can be prefetched and its dependencies polled
// ahead of time; a shared queue was measured slower -- the pop is an exposed L2 round trip and
// early pops reorder the items).  The tiles of task q+1 come BEFORE the rows of task q: while
// some workgroups still finish tiles of a task, the others already transform the previous
// task's rows, and nobody waits at a barrier.  Dependencies are per-task completion counters:
//   B(q) items start when all 32 A(q) items have stored their tiles          (doneA[q])
//   A(q) items store their tile when all B(q-2) items have read that buffer  (doneB[q-2];
//        two mid buffers per team, q mod 2)
//   B(q) items publish (HH row sums / Zdb,Zdr) when all B(q-1) items have    (doneB[q-1];
//        one HH row-sum table per team, written by even tasks, read by odd ones)
// Every dependency points to an EARLIER group and every workgroup walks the groups in order, so
// the workgroup with the earliest unfinished item can always run: no cycle.  Latency hiding:
//   * the counter an item will need is read AHEAD (one lane, the value is looked at a phase
//     later); only if it was not yet satisfied does the workgroup fall into the polling loop;
//   * an item's completion is signalled LATE: its stores drain while the next item's first phase
//     computes, and the count is added at that phase's barrier -- except that a workgroup never
//     enters a polling loop with an unsent completion (it flushes first), which keeps the
//     no-cycle argument valid;
//   * the next tile is requested one phase into the current item (behind a B item's own row
//     loads), so HBM requests are always in flight.
// All spins are bounded and a timeout is reported.  Tiles are stored with plain stores (lines
// stay in the XCD's L2) and rows are read with sc1 loads (bypass the reader's L1).
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

constexpr int FUSED_RING = 8;   // completion counters are a ring over tasks (at most 3 tasks are in flight)
struct FusedCtl {               // zeroed by hipMemsetAsync before every launch; every counter on its own 64-byte line
    unsigned census[8];         // workgroups per XCC
    unsigned arrived;           // grid-wide start counter
    unsigned timeout;           // != 0: a bounded spin gave up
    unsigned pad[6];
    unsigned doneA[8][FUSED_RING][16];      // tiles stored, task q -> slot q % RING, target 32 * (q / RING + 1)
    unsigned doneB[8][FUSED_RING][16];      // row groups finished
};
typedef RangeTile<16> FT;
constexpr int FUSED_THREADS = 1024;
constexpr int FUSED_WAVES = 16;
constexpr int FUSED_ITEMS = 32;                                            // A-items = B-items per task
constexpr int FUSED_STAMP_TASKS = 16;
constexpr int FUSED_OFF_CTL = FT::LDS_BYTES;                               // int [16]
constexpr int FUSED_OFF_TWN = FUSED_OFF_CTL + 64;                          // float2 [512] exp(+2 pi i k / 512)
constexpr int FUSED_LDS_BYTES = FUSED_OFF_TWN + DP_N * 8;                  // 155712 <= 160 KiB
static_assert(FUSED_LDS_BYTES <= 160 * 1024, "fused launch exceeds the CU's LDS");
static_assert(FUSED_WAVES * DP_ELEMS * 8 <= FT::TW_BLK0 * FT::BLK_BYTES, "row buffers must stay below the twiddle pads");
constexpr size_t FUSED_MID_ELEMS = (size_t)(RP_M / 2) * DP_N;              // one channel
// per team: two mid buffers + one HH row-sum table
constexpr size_t FUSED_TEAM_ELEMS = 2 * FUSED_MID_ELEMS + (RP_M / 2) / 2;  // in float2 units

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

template <class Hook0, class Hook>
__device__ __forceinline__ void fused_stage12(unsigned char *smem, float4 (&v)[8], float2 wdv, Hook0 before_barrier1,
                                              Hook after_stage1)
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
    before_barrier1();
    __syncthreads();
    after_stage1();   // v has been consumed: the next tile may be requested into it
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

// group g of a team's list -> (is_B, task q) for T tasks; see the header for the order
__device__ __forceinline__ void fused_decode(int g, int T, bool &isB, int &q)
{
    if (g == 0) { isB = false; q = 0; }
    else if (g == 2 * T - 1) { isB = true; q = T - 1; }
    else if (g & 1) { isB = false; q = (g + 1) / 2; }
    else { isB = true; q = g / 2 - 1; }
}

template <int TAPS>
__global__ __launch_bounds__(FUSED_THREADS) void fused_sector_1024x512(
    const float2 *__restrict__ iq,   // [S][C][1024][512]
    float *__restrict__ out,         // [S][512][2]
    float2 *pool,                    // [8][FUSED_TEAM_ELEMS] per team: mid[2][512][512] + hh[512]
    FusedCtl *ctl, RangeConsts rc, const float2 *__restrict__ tw_n, int n_sectors, int channels, MaTaps taps,
    float k_rr, float k_cal, unsigned long long *stamps /* diagnostics: [grid][FUSED_STAMP_TASKS][8] or nullptr */)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    volatile int *s_ctl = reinterpret_cast<volatile int *>(smem + FUSED_OFF_CTL);
    float2 *s_twn = reinterpret_cast<float2 *>(smem + FUSED_OFF_TWN);
    const int tid = threadIdx.x, w = wave_id(), l = tid & 63;
    const int n = DP_N, gates = RP_M / 2;
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
            if (x == s_ctl[1]) s_ctl[7] = (int)c;
        }
        s_ctl[4] = teams;
        s_ctl[5] = trank;
    }
    __syncthreads();
    // wave-uniform by construction; readfirstlane tells the compiler so (scalar address arithmetic)
    const int xcc = __builtin_amdgcn_readfirstlane(s_ctl[1]);
    const int teams = __builtin_amdgcn_readfirstlane(s_ctl[4]), trank = __builtin_amdgcn_readfirstlane(s_ctl[5]);
    const int T = 2 * ((n_sectors - trank + teams - 1) / teams);             // channel-tasks of this team
    const int rank = __builtin_amdgcn_readfirstlane(s_ctl[2]), sz = __builtin_amdgcn_readfirstlane(s_ctl[7]);
    const int groups = 2 * T;
    float2 *team_pool = pool + (size_t)xcc * FUSED_TEAM_ELEMS;
    float *hh = reinterpret_cast<float *>(team_pool + 2 * FUSED_MID_ELEMS);   // [512] HH row sums of the sector in flight
    float2 *wbuf = reinterpret_cast<float2 *>(smem) + (size_t)w * DP_ELEMS;   // aliases image blocks 0..63 (B items only)
    if (rank >= FUSED_ITEMS) return;   // a team larger than a group: the surplus members own no item

    auto tile_src = [&](int q) { return iq + ((size_t)(trank + (q >> 1) * teams) * channels + (q & 1)) * RP_M * (size_t)n; };
    auto counter = [&](unsigned (*arr)[FUSED_RING][16], int q) { return &arr[xcc][q % FUSED_RING][0]; };
    auto target = [&](int q) { return (unsigned)(FUSED_ITEMS * (q / FUSED_RING + 1)); };
    auto peek = [&](unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };

    unsigned *pend = nullptr;   // completion counter of the previous item: its stores are still draining
    // stores drained by every wave, then one lane counts.  Call sites sit where the drain is free
    // (a phase of compute after the stores) or in front of a polling loop.
    auto flush = [&]() {
        if (pend) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(pend, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pend = nullptr;
        }
    };
    int chk = 0;
    // dependency `*p >= tgt`, whose value lane 0 read a phase ago (`early`): one barrier when it was
    // already satisfied, otherwise flush the unsent completion and poll
    auto resolve = [&](unsigned early, unsigned *p, unsigned tgt) -> bool {
        volatile int *slot = s_ctl + 8 + (chk++ & 3);
        if (tid == 0) *slot = early >= tgt;
        __syncthreads();
        if (*slot) return true;
        flush();
        return team_wait_ge(p, tgt, &ctl->timeout, &s_ctl[0]);
    };

    float4 v[8];         // this lane's share of one range tile (current A item or the prefetched next one)
    float2 wdv;
    bool have = false;   // v holds the tile of the item about to be processed
    unsigned dep1_early = 0;   // lane 0: the next item's first dependency, read ahead
    int g = 0, j = rank;
    int item_no = 0;
#pragma unroll 1
    while (g < groups) {
        bool isB; int q;
        fused_decode(g, T, isB, q);
        // the item after this one (static schedule)
        int ng = g, nj = j + sz;
        if (nj >= FUSED_ITEMS) { ng = g + 1; nj = rank; }
        bool nB = true; int nq = 0;
        if (ng < groups) fused_decode(ng, T, nB, nq);
        // request the next item's tile if it is an A item (zero-record descriptor otherwise: no branch around loads)
        auto prefetch_next = [&]() {
            const bool nv = ng < groups && !nB;
            fused_tile_load(tile_src(nv ? nq : 0), n, (nv ? nj : 0) * 16, rc.wd, v, wdv, nv);
            have = nv;
        };
        // read the next item's first dependency ahead of time (B items only: all tiles of its task stored)
        auto peek_next = [&]() {
            dep1_early = 0;
            if (tid == 0 && ng < groups && nB) dep1_early = peek(counter(ctl->doneA, nq));
        };
        unsigned dep2_early = 0;
        if (stamps && tid == 0 && item_no < FUSED_STAMP_TASKS) {
            unsigned long long *st = stamps + ((size_t)blockIdx.x * FUSED_STAMP_TASKS + item_no) * 8;
            st[0] = __builtin_amdgcn_s_memrealtime(); st[5] = (isB ? 1000u : 0u) + (unsigned)q; st[6] = __builtin_amdgcn_s_memtime();
        }
        if (!isB) {
            // ---------------- A item: range tile j of task q -> mid[q & 1] ----------------
            if (!have) fused_tile_load(tile_src(q), n, j * 16, rc.wd, v, wdv, true);
            have = false;
            const float2 wcur = wdv;
            if (stamps) {   // diagnostics only: when did the tile arrive
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (tid == 0 && item_no < FUSED_STAMP_TASKS)
                    stamps[((size_t)blockIdx.x * FUSED_STAMP_TASKS + item_no) * 8 + 7] = __builtin_amdgcn_s_memrealtime();
            }
            fused_stage12(
                smem, v, wcur,
                [&]() {   // before the barrier after stage 1: the previous item's stores have drained behind the compute
                    if (pend) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                },
                [&]() {   // after it: count the previous item, request the next tile, read this item's dependency ahead
                    if (pend && tid == 0) __hip_atomic_fetch_add(pend, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    pend = nullptr;
                    // the counter read goes first: a wave's loads return in order, behind the tile it would wait for HBM
                    if (q >= 2 && tid == 0) dep2_early = peek(counter(ctl->doneB, q - 2));
                    prefetch_next();
                });
            if (stamps && tid == 0 && item_no < FUSED_STAMP_TASKS)
                stamps[((size_t)blockIdx.x * FUSED_STAMP_TASKS + item_no) * 8 + 1] = __builtin_amdgcn_s_memrealtime();
            // the buffer's previous rows (task q-2) must all have been read before they are overwritten
            if (q >= 2 && !resolve(dep2_early, counter(ctl->doneB, q - 2), target(q - 2))) return;
            if (stamps && tid == 0 && item_no < FUSED_STAMP_TASKS)
                stamps[((size_t)blockIdx.x * FUSED_STAMP_TASKS + item_no) * 8 + 2] = __builtin_amdgcn_s_memrealtime();
            peek_next();
            fused_stage3(smem, team_pool + (size_t)(q & 1) * FUSED_MID_ELEMS, n, j * 16);
            if (stamps && tid == 0 && item_no < FUSED_STAMP_TASKS)
                stamps[((size_t)blockIdx.x * FUSED_STAMP_TASKS + item_no) * 8 + 3] = __builtin_amdgcn_s_memrealtime();
            pend = counter(ctl->doneA, q);   // counted once the stores have drained (see flush)
        } else {
            // ---------------- B item: gates 16 j .. 16 j + 15 of task q, one per wave ------
            if (!resolve(dep1_early, counter(ctl->doneA, q), target(q))) return;
            if (stamps && tid == 0 && item_no < FUSED_STAMP_TASKS)
                stamps[((size_t)blockIdx.x * FUSED_STAMP_TASKS + item_no) * 8 + 1] = __builtin_amdgcn_s_memrealtime();
            const int gate = j * FUSED_WAVES + w;
            cf x[8];
            doppler_load_row<true>(team_pool + (size_t)(q & 1) * FUSED_MID_ELEMS + (size_t)gate * n, l, x);
            if (q >= 1 && tid == 0) dep2_early = peek(counter(ctl->doneB, q - 1));
            // the CU's memory pipeline serves requests in order: no wave's tile request (an HBM miss) may be
            // queued in front of another wave's row loads (L2 hits) -- measured 5 us per item otherwise
            __syncthreads();
            prefetch_next();
            const float s = doppler_row<false, TAPS>(x, wbuf, s_twn, taps, l, gate, false, nodump);
            if (stamps && tid == 0 && item_no < FUSED_STAMP_TASKS)
                stamps[((size_t)blockIdx.x * FUSED_STAMP_TASKS + item_no) * 8 + 2] = __builtin_amdgcn_s_memrealtime();
            // the previous item's stores drained a row transform ago: count it at this barrier
            if (pend) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            volatile int *slot = s_ctl + 8 + (chk++ & 3);
            if (tid == 0) *slot = q < 1 || dep2_early >= target(q - 1);
            __syncthreads();
            if (pend && tid == 0) __hip_atomic_fetch_add(pend, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pend = nullptr;
            if (stamps && tid == 0 && item_no < FUSED_STAMP_TASKS)   // all 16 rows done
                stamps[((size_t)blockIdx.x * FUSED_STAMP_TASKS + item_no) * 8 + 7] = __builtin_amdgcn_s_memrealtime();
            // publish in task order: the HH table is written by even tasks and read by the odd task that follows
            if (!*slot && !team_wait_ge(counter(ctl->doneB, q - 1), target(q - 1), &ctl->timeout, &s_ctl[0])) return;
            if (stamps && tid == 0 && item_no < FUSED_STAMP_TASKS)
                stamps[((size_t)blockIdx.x * FUSED_STAMP_TASKS + item_no) * 8 + 3] = __builtin_amdgcn_s_memrealtime();
            peek_next();
            if (l == 0) {
                if ((q & 1) == 0) {
                    __hip_atomic_store(reinterpret_cast<unsigned *>(&hh[gate]), __builtin_bit_cast(unsigned, s), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    const float shh = __builtin_bit_cast(float, __hip_atomic_load(reinterpret_cast<unsigned *>(&hh[gate]),
                                                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    const int sec = trank + (q >> 1) * teams;
                    reflectivity_store(&out[((size_t)sec * gates + gate) * 2], gate, shh, s, k_rr, k_cal);
                }
            }
            pend = counter(ctl->doneB, q);
        }
        if (stamps && tid == 0 && item_no < FUSED_STAMP_TASKS)
            stamps[((size_t)blockIdx.x * FUSED_STAMP_TASKS + item_no) * 8 + 4] = __builtin_amdgcn_s_memrealtime();
        __syncthreads();   // the LDS image / row buffers are free for the next item
        g = ng;
        j = nj;
        item_no++;
    }
    flush();
}

} // namespace wrp
