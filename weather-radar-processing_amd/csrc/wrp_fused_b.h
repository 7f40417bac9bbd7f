// wrp_fused_b.h -- BASELINE configs[4] (m = 2048 range cells x n = 128 pulses) in ONE persistent launch: the team
// protocol of wrp_fused.h (XCD teams of 32 tile + 32 row workgroups, one 1 MiB hand-over slot per XCD that never
// leaves its L2, byte flags, bounded waits, self-zeroing control block) around the arithmetic of wrp_shape_b.h.
// The reference cannot run this shape at all (rpv2.cu:84,145-148); the two-kernel form (range_pass_2048 +
// doppler_pass_128) moves the 4 MiB intermediate of a sector out to HBM and back: 2.0 x the algorithmic bytes.
//
// What differs from the 1024 x 512 launch:
//   * a sector is ONE task: its two channels are transformed side by side -- tile member t takes channel t >> 4 and
//     the 8-COLUMN tile t & 15 (2048 rows x 64 bytes = the same 128 KiB of registers per workgroup as 1024 rows x
//     128 bytes; the two members t, t ^ 1 ask for the two halves of every 128-byte line at the same time);
//   * 2048 = 16 x 16 x 8: stage 1 (registers) radix 16 over rows p0 + 128 r, stage 2 (LDS, wave-local) radix 16 over
//     positions p1 + 8 r of the sub-transform the wave owns, stage 3 radix 8; the sixteen 128-point sub-transforms go
//     through LDS in two groups of eight as there, the gates of group g are those with (gate mod 16) in [8 g, 8 g + 8):
//     half g of the slot (layout: fused_b_store);
//   * LDS: the group image [8 k1][16 blocks of 8 positions x 64 B + 64 B of padding] is 72 KiB as there; its 128 pads
//     hold the stage-1 twiddles W_2048^{p0 k1} for k1 = 1 .. 8 only (one pad per p0) -- k1 = 9 .. 15 are the products
//     W^{8 p0} W^{(k1 - 8) p0} (seven complex multiplies per lane and tile; range_pass_2048 forms them the same way,
//     so the two forms stay bit-identical) -- and the range window is kept as its first half (it is symmetric: the
//     engine stores wr_c[i] = wr_c[m - 1 - i] exactly): 78,912 bytes, two workgroups per CU;
//   * a row wave transforms FOUR rows at a time (16 lanes per row, 128 = 8 x 4 x 4: doppler_row_128), the HH and the
//     VV row of two gates, so Zdb and Zdr leave with the task that produced them; every wave serves both halves;
//   * wire-format input (RAW instantiations): the tile members read the 12-byte samples -- or the 8-byte ones of
//     WRP_FLAG_WIRE_8 -- themselves: ONE 16-byte load per lane and row as in the planar form (fused_b_tile_addr), byte swap +
//     conversion in stage 1.
#pragma once
#include <hip/hip_runtime.h>

#include "wrp_fused.h"
#include "wrp_shape_b.h"

namespace wrp {

struct FusedTileB {
    static constexpr int ROW_BYTES = 64, BLK_BYTES = 9 * ROW_BYTES, BLOCKS = 128;
    static constexpr int IMG_BYTES = BLOCKS * BLK_BYTES;          // 73728
    static constexpr int OFF_WR = IMG_BYTES;                      // float wr_c[1024]: the first half of the window
    static constexpr int OFF_CTL = OFF_WR + (RB_M / 2) * 4;       // 77824: control words, both kinds (as FusedTile)
    static constexpr int OFF_TW2 = OFF_CTL + 64;                  // float2 [16 k2][8 p1]: W_128^{p1 k2}
    static constexpr int OFF_STAMPS = OFF_TW2 + 16 * 8 * 8;       // diagnostics instantiation (tools/fused_stamps_b.py): [16 tasks][9] 64-bit stamps
    static constexpr int LDS_BYTES = OFF_STAMPS + FUSED_STAMP_TASKS * FUSED_STAMPS * 8;   // 80064: two workgroups per CU
    static constexpr int OFF_TWN = 8 * 4 * DB_ROW_ELEMS * 8;      // row workgroup: 8 waves x 4 row buffers, then exp(+2 pi i k / 128)
    static_assert(OFF_TWN + RB_N * 8 <= OFF_CTL, "row workgroup layout fits");
    static_assert(OFF_CTL == FusedTile::OFF_CTL, "the control words sit where fused_join / fused_leave expect them");
    static_assert(2 * LDS_BYTES <= 160 * 1024 && 3 * LDS_BYTES > 160 * 1024, "exactly two workgroups per CU");
    // position pos = k1l * 128 + p of the group image, column pair cp (16 bytes)
    static __device__ __forceinline__ int addr(int pos, int cp) { return (pos >> 3) * BLK_BYTES + (pos & 7) * ROW_BYTES + cp * 16; }
    // In the rows of the positions with bit 1 set the two columns of every pair are swapped (column c sits at (c ^ swz(pos)) * 8).
    // Stage 1 writes ONE column of a pair at a time, 8 bytes per lane at a 16-byte stride, and a ds_write_b64 serves sixteen
    // lanes = four 64-byte rows at once: unswapped, the pieces of rows p and p + 2 fall on the same banks (two passes per
    // write: tools/lds_banks.py model, 256 against 128 LDS cycles per wave and column pair); swapped, they interleave.  Every
    // other access covers whole rows and only sees its columns permuted (FusedTile::swz is the same idea at 128-byte rows).
    static __device__ __forceinline__ int swz(int pos) { return (pos >> 1) & 1; }
    // W_2048^{p0 (j + 1)}, j < 8: the eight entries of lane p0 are the 64 bytes of pad p0
    static __device__ __forceinline__ int tw1_addr(int p0, int j) { return p0 * BLK_BYTES + 8 * ROW_BYTES + j * 8; }
    static __device__ __forceinline__ int tw2_addr(int p1, int k2) { return OFF_TW2 + (k2 * 8 + p1) * 8; }
};

// the fifteen stage-1 twiddles of a lane from the eight that are stored (shared with range_pass_2048: same products)
__device__ __forceinline__ void rb_derive_twiddles(cf (&tw)[16])
{
#pragma unroll
    for (int j = 1; j < 8; j++) tw[8 + j] = cmul(tw[8], tw[j]);
}

// a QUARTER of the lane's 16 row loads (rows p0 + 128 r, r = QUARTER mod 4).  The neighbouring member asks for the other
// half of every 128-byte line at the same time and finds it in (or on its way into) the L2.  Non-temporal like the
// 1024 x 512 launch's input: beside a PLAIN stream the rewritten slot does not stay in the L2 (83 % of the intermediate
// was written back: WRITE_SIZE 1.75 MB per sector, profiles/r03/fused_b_input_policy.log).
constexpr int FUSED_B_INPUT_AUX = AUX_NT;
// the row waves touch the slow lines (wrp_fused.h: fused_touch_slow_lines) of the input of the task this many tasks ahead
// (the tile members request a task's input during the task before it), behind their hand-over of this half; 0 = never.
// Measured (profiles/r05/ab_b_slow_line_touch.log): 2 / half 0 -14.2 %, 2 / half 1 -12.6 %, 3 -12.5 %, 1 +6.6 %
constexpr int FUSED_B_TOUCH_AHEAD = 2, FUSED_B_TOUCH_HALF = 0;
// Request pacing (wrp_fused.h): 16 = one load at a time over the task, 1.69 us/sector, HBM traffic 1.13 x the algorithmic
// bytes; 4 = four quarters, 1.74 us/sector, 1.06 x (the smoother stream leaves fewer non-temporal lines per L2 set to evict
// in place of the slot's): profiles/r03/fused_b_input_policy.log, ab_request_pacing_B.log.
// RAW (wire-format input): src = the SECTOR's 12-byte samples (hhI hhQ vvI vvQ vhI vhQ, big-endian int16,
// sector.cpp:52-62) + 4 ch bytes; the lane's two samples of a row are 24 contiguous bytes of which the 16 from byte 4 ch
// on hold its channel's dword of the first sample in .x and of the second in .w (.y, .z: the dwords in between, never
// used) -- ONE 16-byte load per row as in the planar form, the same sixteen requests per task, no decode pass.
// RAW = 8 (hhI hhQ vvI vvQ: the feeder has dropped VH): the lane's two samples ARE 16 contiguous bytes, hh0 vv0 hh1 vv1; both
// channels' members load the same bytes and pick their dword in the byte permute of the conversion (wire_sample2) -- the
// planar form's geometry (64 bytes of every 1 KiB row per member) with half the bytes from HBM.
template <int RAW>
__device__ __forceinline__ void fused_b_tile_addr(const float2 *src, int col_base, bool valid, rsrc_t &rs, int &voff, int &row_stride)
{
    const int w = wave_id();
    int l = threadIdx.x & 63;
    asm volatile("" : "+v"(l));
    const int p0 = w * 16 + (l >> 2), cp = l & 3;
    rs = make_rsrc(src, valid ? (unsigned)RB_M * RB_N * (RAW ? RAW : 8) : 0u);
    voff = (p0 * RB_N + col_base + cp * 2) * (RAW ? RAW : 8);
    row_stride = RB_N * (RAW ? RAW : 8);
}
template <int QUARTER, int RAW = 0>
__device__ __forceinline__ void fused_b_tile_load(const float2 *src /* wave-uniform */, int col_base, const float *wd,
                                                  float4 (&v)[16], float2 &wdv, bool valid)
{
    int l = threadIdx.x & 63;
    asm volatile("" : "+v"(l));
    const int cp = l & 3;
    rsrc_t rs;
    int voff, row_stride;
    fused_b_tile_addr<RAW>(src, col_base, valid, rs, voff, row_stride);
#pragma unroll
    for (int r = QUARTER; r < 16; r += 4) v[r] = buf_load_f4<FUSED_B_INPUT_AUX>(rs, voff, 128 * r * row_stride);
    if (QUARTER == 3) wdv = buf_load_f2<0>(make_rsrc(wd, (unsigned)RB_N * 4u), (col_base + cp * 2) * 4, 0);
}

// ONE row load of the next tile (rows p0 + 128 R): the tile is requested a piece at a time over the task, see wrp_fused.h
template <int R, int RAW = 0>
__device__ __forceinline__ void fused_b_tile_load1(const float2 *src /* wave-uniform */, int col_base, const float *wd,
                                                   float4 (&v)[16], float2 &wdv, bool valid)
{
    int l = threadIdx.x & 63;
    asm volatile("" : "+v"(l));
    const int cp = l & 3;
    rsrc_t rs;
    int voff, row_stride;
    fused_b_tile_addr<RAW>(src, col_base, valid, rs, voff, row_stride);
    v[R] = buf_load_f4<FUSED_B_INPUT_AUX>(rs, voff, 128 * R * row_stride);
    if (R == 15) wdv = buf_load_f2<0>(make_rsrc(wd, (unsigned)RB_N * 4u), (col_base + cp * 2) * 4, 0);
}

__device__ __forceinline__ void fused_b_stage1_tables(const unsigned char *smem, cf (&tw)[16])
{
    typedef FusedTileB T;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    tid &= FUSED_THREADS - 1;
    const int p0 = (tid >> 6) * 16 + ((tid & 63) >> 2);
    const unsigned char *pad = smem + T::tw1_addr(p0, 0);
#pragma unroll
    for (int j = 0; j < 8; j++) tw[j + 1] = *reinterpret_cast<const float2 *>(pad + j * 8);
    rb_derive_twiddles(tw);
}

template <int COLUMN, int RAW = 0>
__device__ __forceinline__ void fused_b_stage1(unsigned char *smem, const float4 (&v)[16], float2 wdv, const cf (&tw)[16], cf (&g)[8],
                                               unsigned wire_sel = 0 /* RAW = 8: WIRE_SEL_HH or WIRE_SEL_VV, wave-uniform */)
{
    typedef FusedTileB T;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    tid &= FUSED_THREADS - 1;
    const int w = tid >> 6, l = tid & 63, cp = l & 3;
    const int p0 = w * 16 + (l >> 2);
    const int slot = T::addr(p0, cp) + 8 * (COLUMN ^ T::swz(p0));   // position k1*128 + p0 is 16 k1 blocks further on (same bit 1)
    const float *s_wr = reinterpret_cast<const float *>(smem + T::OFF_WR);
    float wr[16];    // rows p0 + 128 r; the second half of the window mirrored: wr_c[i] = wr_c[2047 - i]
#pragma unroll
    for (int r = 0; r < 16; r++) wr[r] = r < 8 ? s_wr[p0 + 128 * r] : s_wr[(RB_M - 1 - 128 * r) - p0];
    cf a[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const float wgt = wr[r] * (COLUMN ? wdv.y : wdv.x);
        if (RAW == 12) a[r] = cscale(wire_sample(COLUMN ? v[r].w : v[r].x), wgt);   // exact conversion: bit-identical to decode_wire + the planar form
        else if (RAW == 8) a[r] = cscale(COLUMN ? wire_sample2(v[r].z, v[r].w, wire_sel) : wire_sample2(v[r].x, v[r].y, wire_sel), wgt);
        else a[r] = COLUMN ? cscale(make_float2(v[r].z, v[r].w), wgt) : cscale(make_float2(v[r].x, v[r].y), wgt);
    }
    fft16<-1>(a);
    *reinterpret_cast<float2 *>(smem + slot) = a[0];
#pragma unroll
    for (int k1 = 1; k1 < 8; k1++) *reinterpret_cast<float2 *>(smem + slot + k1 * 16 * T::BLK_BYTES) = cmul(a[k1], tw[k1]);
#pragma unroll
    for (int k1 = 8; k1 < 16; k1++) g[k1 - 8] = cmul(a[k1], tw[k1]);
}

__device__ __forceinline__ void fused_b_group1_to_lds(unsigned char *smem, const cf (&ga)[8], const cf (&gc)[8])
{
    typedef FusedTileB T;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    tid &= FUSED_THREADS - 1;
    const int w = tid >> 6, l = tid & 63, cp = l & 3;
    const int p0 = w * 16 + (l >> 2);
    const int first = T::addr(p0, cp) + 8 * T::swz(p0), second = T::addr(p0, cp) + 8 * (1 ^ T::swz(p0));   // two b64 = one b128 in LDS cycles
#pragma unroll
    for (int j = 0; j < 8; j++) {
        *reinterpret_cast<float2 *>(smem + first + j * 16 * T::BLK_BYTES) = ga[j];
        *reinterpret_cast<float2 *>(smem + second + j * 16 * T::BLK_BYTES) = gc[j];
    }
}

// stage 2 of the sub-transform this wave owns (image blocks w*16 ..): ONE radix-16 item per lane -- column l & 7,
// positions p1 + 8 r with p1 = l >> 3; a wave-instruction covers 8 positions x 64 contiguous bytes: no bank conflict
__device__ __forceinline__ void fused_b_stage2(unsigned char *smem)
{
    typedef FusedTileB T;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    tid &= FUSED_THREADS - 1;
    const int w = tid >> 6, l = tid & 63, col = l & 7, p1 = l >> 3;
    unsigned char *base = smem + w * 16 * T::BLK_BYTES + p1 * T::ROW_BYTES + (col ^ T::swz(p1)) * 8;   // positions p1 + 8 r: all of p1's bit 1
    cf a[16], t[16];
#pragma unroll
    for (int r = 0; r < 16; r++) a[r] = *reinterpret_cast<const float2 *>(base + r * T::BLK_BYTES);
#pragma unroll
    for (int k2 = 1; k2 < 16; k2++) t[k2] = *reinterpret_cast<const float2 *>(smem + T::tw2_addr(p1, k2));
    fft16<-1>(a);
    *reinterpret_cast<float2 *>(base) = a[0];
#pragma unroll
    for (int k2 = 1; k2 < 16; k2++) *reinterpret_cast<float2 *>(base + k2 * T::BLK_BYTES) = cmul(a[k2], t[k2]);
    wave_lds_fence();
}
// stage 3: two radix-8 items per lane, k2 = (l >> 3) + 8 it; the 64-byte pad after every 8 positions spreads the eight
// k2 of a wave-instruction over the banks.  Item `it` yields the gates k1 + 16 k2 + 256 k3, k3 < 4 (the rest is never read).
__device__ __forceinline__ void fused_b_stage3(unsigned char *smem, cf (&o)[2][4])
{
    typedef FusedTileB T;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    tid &= FUSED_THREADS - 1;
    const int w = tid >> 6, l = tid & 63, col = l & 7;
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const int k2 = (l >> 3) + 8 * it;
        const unsigned char *base = smem + (w * 16 + k2) * T::BLK_BYTES + col * 8, *base_swz = smem + (w * 16 + k2) * T::BLK_BYTES + (col ^ 1) * 8;
        cf a[8];
#pragma unroll
        for (int r = 0; r < 8; r++) a[r] = *reinterpret_cast<const float2 *>((T::swz(r) ? base_swz : base) + r * T::ROW_BYTES);   // position k2*8 + r
        fft8<-1>(a);
#pragma unroll
        for (int k3 = 0; k3 < 4; k3++) o[it][k3] = a[k3];
    }
}
// The slot of shape B: [2 channels][256 pair rows Q][16 tiles][2 gates][8 columns] complex.  A tile member holds only 8
// columns = 64 bytes of a gate's row, and a store that covers half a 128-byte line is written THROUGH by the L2 (first
// form, one gate per slot row: 83 % of the intermediate went out to HBM, 1.75 MB per sector of WRITE_SIZE): so the two
// gates k2 = 2 a, 2 a + 1 of a store instruction's lane pairs share a line -- 16 lanes x 8 bytes, whole lines only -- and
// a row wave reads both of them (and both channels) in the same pass.  Pair row Q of half g:
//   Q = ((k3 * 2 + it) * 4 + (k2l >> 1)) * 8 + (k1 - 8 g)      with k2 = k2l + 8 it, k2l < 8
//   gate(Q, pb) = 8 g + (Q & 7) + 16 * (2 * ((Q >> 3) & 3) + pb + 8 * ((Q >> 5) & 1)) + 256 * (Q >> 6),   pb = k2l & 1
__device__ __forceinline__ int fused_b_gate(int g, int Q, int pb)
{
    return 8 * g + (Q & 7) + 16 * (2 * ((Q >> 3) & 3) + pb + 8 * ((Q >> 5) & 1)) + 256 * (Q >> 6);
}
// tee (diagnostics instantiation only): the sector-channel's [1024 gates][128] block of a copy of the intermediate, g: the half
__device__ __forceinline__ void fused_b_store(float2 *mid /* wave-uniform */, int ch, int col_base, const cf (&o)[2][4], float2 *tee = nullptr, int g = 0)
{
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    tid &= FUSED_THREADS - 1;
    const int w = tid >> 6, l = tid & 63, c8 = l & 7, k2l = l >> 3;
    const rsrc_t rd = make_rsrc(mid, (unsigned)FUSED_TEAM_ELEMS * 8u);
    const int voff = ((ch * 256 + (k2l >> 1) * 8 + w) * 16 + (col_base >> 3)) * 128 + (k2l & 1) * 64 + c8 * 8;
#pragma unroll
    for (int it = 0; it < 2; it++)
#pragma unroll
        for (int k3 = 0; k3 < 4; k3++) {   // row offset in the VGPR, soffset 0: see buf_store_f4
            v2f t;
            t.x = o[it][k3].x; t.y = o[it][k3].y;
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, t), rd, voff + (k3 * 2 + it) * 4 * 8 * 16 * 128, 0, 0);
        }
    if (tee) {
#pragma unroll
        for (int it = 0; it < 2; it++)
#pragma unroll
            for (int k3 = 0; k3 < 4; k3++)
                tee[(size_t)fused_b_gate(g, ((k3 * 2 + it) * 4 + (k2l >> 1)) * 8 + w, k2l & 1) * RB_N + col_base + c8] = o[it][k3];
    }
}

template <int TAPS, bool STAMPS = false, bool TEE = false, int RAW = 0 /* wire-format input: bytes per sample (12 or 8), 0 = planar */>
__global__ __launch_bounds__(FUSED_THREADS, 4) __attribute__((amdgpu_waves_per_eu(4, 4))) void fused_chain_2048x128(
    const float2 *__restrict__ iq,   // [S][C][2048][128]; RAW: the wire format, [S][2048 x 128][RAW bytes]
    float *__restrict__ out,         // [S][1024][2]
    float2 *pool,                    // [8][FUSED_TEAM_ELEMS]: per team ONE slot [2 channels][512][128] through which both halves go
    FusedCtl *ctl, RangeConsts rc /* wr_c symmetric */, const float2 *__restrict__ tw_n /* exp(+2 pi i k / 128) */, int n_sectors,
    int channels, MaTaps taps, float k_rr, float k_cal, unsigned *host_status, unsigned long long *stamps /* STAMPS instantiation: [grid][FUSED_STAMP_TASKS][9] */,
    unsigned *frames /* optional (N2): [S][2][1 + 1024] words */, const unsigned *frame_hdrs /* [S] header words */,
    float2 *tee /* diagnostics (TEE instantiation): [S][C][1024][128], a copy of everything that goes through the slots */)
{
    typedef FusedTileB T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_word *s_ctl = (lds_word *)(smem + T::OFF_CTL);
    const int tid = threadIdx.x, w = wave_id(), l = tid & 63;
    const int gates = RB_M / 2;

    const FusedSeat seat = fused_join(ctl, s_ctl);
    const int xcc = seat.xcc, kind = seat.kind, rank = seat.rank, teams = seat.teams, trank = seat.trank;
    if (!seat.ok || rank >= FUSED_MEMBERS) {
        fused_leave(ctl, host_status, xcc, s_ctl);
        return;
    }
    const int tasks = (n_sectors - trank + teams - 1) / teams;   // sectors of this team: one task each
    float2 *mid = pool + (size_t)xcc * FUSED_TEAM_ELEMS;
    // diagnostics instantiation only: stamps go to LDS and are copied out at the end (tools/fused_stamps_b.py)
    unsigned long long *s_stamps = reinterpret_cast<unsigned long long *>(smem + T::OFF_STAMPS);
    if (STAMPS) {
        for (int e = tid; e < FUSED_STAMP_TASKS * FUSED_STAMPS; e += FUSED_THREADS) s_stamps[e] = 0;
        __syncthreads();
        if (tid == 0) s_stamps[FUSED_STAMPS - 1] = ((unsigned long long)kind << 32) | ((unsigned long long)xcc << 16) | (unsigned)rank;
    }
    const unsigned long long clk_t0 = STAMPS ? __builtin_amdgcn_s_memtime() : 0, clk_r0 = STAMPS ? __builtin_amdgcn_s_memrealtime() : 0;
    auto stamp = [&](int q, int k) {   // wave 0 of the workgroup, slots 0 .. 7 of the task
        if (STAMPS && w == 0 && q < FUSED_STAMP_TASKS) {
            const unsigned long long t = __builtin_amdgcn_s_memrealtime();
            if (l == 0) s_stamps[q * FUSED_STAMPS + k] = t;
        }
    };
    auto flush_stamps = [&]() {
        if (STAMPS && stamps) {
            if (tid == 0) {
                s_stamps[1 * FUSED_STAMPS + FUSED_STAMPS - 1] = __builtin_amdgcn_s_memtime() - clk_t0;
                s_stamps[2 * FUSED_STAMPS + FUSED_STAMPS - 1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
            }
            __syncthreads();
            for (int e = tid; e < FUSED_STAMP_TASKS * FUSED_STAMPS; e += FUSED_THREADS)
                stamps[(size_t)blockIdx.x * FUSED_STAMP_TASKS * FUSED_STAMPS + e] = s_stamps[e];
        }
    };

    if (kind == 0) {
        // =============================== tile member: channel rank >> 4, 8-column tile rank & 15 ===============================
        const int ch = rank >> 4;
        const unsigned wire_sel = __builtin_amdgcn_readfirstlane(ch ? WIRE_SEL_VV : WIRE_SEL_HH);
        // the member's 8-column tile of task q: rotated by TWO from task to task (the pair t, t ^ 1 keeps asking for the two
        // halves of the same lines), so that a tile that loads slowly is a transient of every member instead of one member
        // that is late in every task (wrp_fused.h: tile_col)
        auto tile_col = [&](int q) { return (((rank & 15) + 2 * q) & 15) * 8; };
        auto tile_src = [&](int q) {   // RAW: the sector's samples from byte 4 ch on (float2 units: 12 bytes = 1.5)
            // 12-byte samples: from byte 4 ch on (the channel's dwords of a lane's two samples are then .x and .w of its 16
            // bytes); 8-byte samples: the lane's two samples ARE its 16 bytes (hh0 vv0 hh1 vv1), the channel is picked in stage 1
            if (RAW) return reinterpret_cast<const float2 *>(reinterpret_cast<const unsigned char *>(iq) + (size_t)(trank + q * teams) * RB_M * RB_N * RAW + (RAW == 12 ? 4 * ch : 0));
            return iq + ((size_t)(trank + q * teams) * channels + ch) * RB_M * (size_t)RB_N;
        };
        float4 v[16];
        float2 wdv;
        fused_b_tile_load<0, RAW>(tile_src(0), tile_col(0), rc.wd, v, wdv, tasks > 0);
        fused_b_tile_load<1, RAW>(tile_src(0), tile_col(0), rc.wd, v, wdv, tasks > 0);
        fused_b_tile_load<2, RAW>(tile_src(0), tile_col(0), rc.wd, v, wdv, tasks > 0);
        fused_b_tile_load<3, RAW>(tile_src(0), tile_col(0), rc.wd, v, wdv, tasks > 0);
        for (int e = tid; e < RB_M / 2; e += FUSED_THREADS) {
            const int p0 = e >> 3, j = e & 7;
            *reinterpret_cast<float2 *>(smem + T::tw1_addr(p0, j)) = rc.tw[(p0 * (j + 1)) & (RB_M - 1)];
            reinterpret_cast<float *>(smem + T::OFF_WR)[e] = rc.wr_c[e];
        }
        if (tid < 128) *reinterpret_cast<float2 *>(smem + T::tw2_addr(tid & 7, tid >> 3)) = rc.tw[(16 * (tid & 7) * (tid >> 3)) & (RB_M - 1)];
        __syncthreads();
        int *s_arrived = reinterpret_cast<int *>(smem + T::OFF_CTL + 56);
        if (tid < 2) s_arrived[tid] = 0;
        __syncthreads();
        const FusedFlags *my_loaded0 = &ctl->loaded[0][xcc][rank], *my_loaded1 = &ctl->loaded[1][xcc][rank];
        int failed = 0;
#pragma unroll 1
        for (int q = 0; q < tasks; q++) {
            cf ga[8], gc[8];
            {
                cf tw[16];
                stamp(q, 0);
                fused_b_stage1_tables(smem, tw);
                fused_b_stage1<0, RAW>(smem, v, wdv, tw, ga, wire_sel);
                stamp(q, 5);
                // half 1 of the previous task was stored half a stage ago: drained and counted here (see wrp_fused.h)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                int last = 0;
                if (l == 0) last = atomicAdd(s_arrived, 1) == 8 * q + 7;
                if (q > 0 && __builtin_amdgcn_readfirstlane(last)) l2_flag32(ctl->stored[1][xcc], l, rank, (unsigned)q);
                fused_b_stage1<1, RAW>(smem, v, wdv, tw, gc, wire_sel);
            }
            __syncthreads();                    // A1: group 0 is in the image
            stamp(q, 1);
            const float2 *next = tile_src(q + 1 < tasks ? q + 1 : 0);
            const bool more = q + 1 < tasks;
            const int col_base = tile_col(q), next_col = tile_col(q + 1);
            cf o[2][4];
#define WRP_LB(R) fused_b_tile_load1<R, RAW>(next, next_col, rc.wd, v, wdv, more)
            WRP_LB(0); WRP_LB(8);
            fused_b_stage2(smem);
            WRP_LB(4); WRP_LB(12);
            fused_b_stage3(smem, o);
            WRP_LB(1); WRP_LB(9);
            spin_flags_sticky(my_loaded1, (unsigned)q, failed, w != 0);     // the slot still holds half 1 of task q - 1
            __syncthreads();                    // A2: group 0 has left the image; the slot is free for half 0
            float2 *tee_task = TEE && tee ? tee + ((size_t)(trank + q * teams) * channels + ch) * (RB_M / 2) * RB_N : nullptr;
            fused_b_store(mid, ch, col_base, o, tee_task, 0);
            __builtin_amdgcn_sched_barrier(0);  // the loads below stay BEHIND the stores: the counted wait tells them apart
            // four requests behind the stores; with 12-byte samples (a member's pieces straddle lines) two, and two more behind
            // A3: -2.8 % there, nothing for the other forms (profiles/r05/ab_b_after_touch_variants.log)
            WRP_LB(5); WRP_LB(13);
            if (RAW != 12) { WRP_LB(2); WRP_LB(10); }
            fused_b_group1_to_lds(smem, ga, gc);
            stamp(q, 2);
            // all but the requests just issued: the stores are in the L2
            if (RAW != 12) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            stamp(q, 6);
            int last = 0;
            if (l == 0) last = atomicAdd(s_arrived + 1, 1) == 8 * q + 7;
            if (__builtin_amdgcn_readfirstlane(last)) l2_flag32(ctl->stored[0][xcc], l, rank, (unsigned)(q + 1));
            __syncthreads();                    // A3: group 1 is in the image
            stamp(q, 3);
            WRP_LB(6);
            if (RAW == 12) { WRP_LB(2); WRP_LB(10); }
            fused_b_stage2(smem);      // (its sixteen points + fifteen twiddles need registers: most of the rest is requested behind it)
            WRP_LB(14); WRP_LB(3);
            fused_b_stage3(smem, o);
            WRP_LB(11); WRP_LB(7); WRP_LB(15);
#undef WRP_LB
            stamp(q, 7);
            spin_flags_sticky(my_loaded0, (unsigned)(q + 1), failed, w != 0);
            stamp(q, 4);
            __syncthreads();                    // A4: image free for the next stage 1; the rows have half 0 of THIS task
            fused_b_store(mid, ch, col_base, o, tee_task, 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tasks > 0 && w == 0) l2_flag32(ctl->stored[1][xcc], l, rank, (unsigned)tasks);
        if (failed && l == 0) __hip_atomic_store(&ctl->status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        flush_stamps();
        fused_leave(ctl, host_status, xcc, s_ctl);
    } else {
        // =============================== row member ===============================
        // Every wave serves both halves: per half ONE pass over pair row Q = 8 rank + w -- four 16-lane rows: the two
        // gates of the pair x {HH, VV} -- whole 128-byte lines per load instruction; Zdb / Zdr of both gates leave at once.
        float2 *s_twn = reinterpret_cast<float2 *>(smem + T::OFF_TWN);
        for (int e = tid; e < RB_N; e += FUSED_THREADS) s_twn[e] = tw_n[e];
        if (tid < 4) s_ctl[12 + tid] = 0;
        __syncthreads();
        const int sub = l >> 4, i = l & 15, pb = sub & 1, chn = sub >> 1;
        float2 *rbuf = reinterpret_cast<float2 *>(smem) + (size_t)(w * 4 + sub) * DB_ROW_ELEMS;
        const rsrc_t rs = make_rsrc(mid, (unsigned)FUSED_TEAM_ELEMS * 8u);
        const int Q = 8 * rank + w;
        const int voff = (chn * 256 + Q) * 2048 + (i >> 3) * 128 + pb * 64 + (i & 7) * 8;   // element j = i + 16 r: tile 2 r + (i >> 3)
        // the slow lines of the sectors to come (wrp_fused.h: fused_touch_slow_lines): the bytes of task q that the tile
        // members read -- planes HH and VV of the planar block, or the sector's wire bytes
        const unsigned touch_bytes = RAW ? (unsigned)RB_M * RB_N * RAW : 2u * RB_M * RB_N * 8u;
        auto touch = [&](int q) {
            const unsigned char *p = reinterpret_cast<const unsigned char *>(iq) +
                (size_t)(trank + (q >= 0 && q < tasks ? q : 0) * teams) * (RAW ? (size_t)RB_M * RB_N * RAW : (size_t)channels * RB_M * RB_N * 8);
            return fused_touch_slow_lines(p, touch_bytes, 8 * rank + w, l, FUSED_B_TOUCH_AHEAD > 0 && q >= 0 && q < tasks);
        };
        float touched = touch(FUSED_B_TOUCH_AHEAD - 1);      // (task 0's tile is on its way already)
#pragma unroll 1
        for (int q = 0; q < tasks; q++) {
            bool there = true;
            const int sec = trank + q * teams;
            float *o2 = &out[(size_t)sec * gates * 2];
            unsigned *fr = frames ? frames + (size_t)sec * 2 * (1 + gates) : nullptr;
#pragma unroll
            for (int g = 0; g < 2; g++) {
                stamp(q, 4 * g);
                there = there && spin_flags(&ctl->stored[g][xcc][rank], (unsigned)(q + 1), &ctl->status);
                if (!there) break;                              // status is set: the launch is void
                stamp(q, 4 * g + 1);
                __builtin_amdgcn_s_setprio(FUSED_ROW_PRIO);     // the notice-to-`loaded` stretch at raised priority: wrp_fused.h
                cf x[8];
#pragma unroll
                for (int r = 0; r < 8; r++) x[r] = buf_load_f2<AUX_SC1>(rs, voff, r * 256);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // rows in registers: the slot may be overwritten
                int last = 0;
                if (l == 0) last = atomicAdd(reinterpret_cast<int *>(smem + T::OFF_CTL + 48 + 4 * g), 1) == 8 * q + 7;
                if (__builtin_amdgcn_readfirstlane(last)) l2_flag32(ctl->loaded[g][xcc], l, rank, (unsigned)(q + 1));
                __builtin_amdgcn_s_setprio(0);
                stamp(q, 4 * g + 2);
                if (FUSED_B_TOUCH_AHEAD > 0 && g == FUSED_B_TOUCH_HALF) {
                    // behind the wave's part of the hand-over chain, a transform and a wait in front of its next row loads
                    // (which return in order behind this one); the previous touch has long landed
                    asm volatile("" :: "v"(touched));
                    touched = touch(q + FUSED_B_TOUCH_AHEAD);
                }
                const int gate = fused_b_gate(g, Q, pb);
                const float S = doppler_row_128<TAPS>(x, rbuf, s_twn, taps, i);
                const float other = __shfl(S, (l + 32) & 63);     // the VV row sum sits 32 lanes above the HH one
                if (i == 0 && chn == 0) reflectivity_store(o2 + 2 * gate, gate, S, other, k_rr, k_cal, fr, gates, fr ? frame_hdrs[sec] : 0u);
                stamp(q, 4 * g + 3);
            }
            if (!there) break;
        }
        asm volatile("" :: "v"(touched));
        flush_stamps();
        fused_leave(ctl, host_status, xcc, s_ctl);
    }
}

} // namespace wrp
