// wrp_shape_b.h -- tuned kernels for BASELINE.json configs[4]: m = 2048 range cells x n = 128 pulses
// (the reference cannot run this shape at all: __clip_v2<<<2, m>>> needs m <= 1024 and d_ma[512],
// rpv2.cu:84,145-148).  Same chain and semantics as wrp_kernels.h.  Stage dumps: the Doppler stages (03 .. rowsum)
// and the half-height intermediate (WRP_STAGE_MID) come from THESE kernels; 01hamm and 02fft1 (all m rows, which
// range_pass_2048 never forms) from the generic kernels of wrp_generic.h, against which these are tested too.
//
//   range_pass_2048   : a2 + a3.  2048 = 16 x 16 x 8 over the positions p of a column, one 1024-thread
//                       workgroup per CU walking 16-column tiles (whole 128-byte lines) with the next
//                       tile requested while the current one is transformed:
//       stage 1 (registers, from HBM): lane owns rows p0 + 128 r, r < 16, of two columns -> radix 16,
//               twiddle W_2048^{p0 k1} -> position k1*128 + p0
//       stage 2 (LDS, one column per lane): positions k1*128 + p1 + 8 r, r < 16 -> radix 16, twiddle
//               W_128^{p1 k2}, in place
//       stage 3 (LDS): positions k1*128 + k2*8 + r, r < 8 -> radix 8; gate k = k1 + 16 k2 + 256 k3, of
//               which k3 < 4 (gates < m/2) are computed and stored (the chain never reads the rest).
//     A whole tile would need 256 KiB of LDS: as in the fused launch the sixteen 128-point
//     sub-transforms go through LDS in two groups of eight k1 (144 KiB image incl. padding, the
//     2048-entry twiddle table in the pads), group 1 waiting in 32 registers.
//   doppler_pass_128  : a4 .. a9.  A row is 1 KiB: 16 lanes per row, 128 = 8 x 4 x 4; one wave owns
//                       two gates x {HH, VV}, so Zdr needs no second kernel and every reduction is a
//                       DPP reduction inside a row of 16 lanes.
#pragma once
#include <hip/hip_runtime.h>

#include "wrp_kernels.h"

namespace wrp {

constexpr int RB_M = 2048, RB_N = 128;

struct RangeTileB {   // image of ONE group: [8 k1][128 positions][16 columns], 9-row blocks as RangeTile<16>
    static constexpr int ROW_BYTES = 128, BLK_BYTES = 9 * ROW_BYTES, BLOCKS = 128, THREADS = 1024;
    static constexpr int IMG_BYTES = BLOCKS * BLK_BYTES;          // 147456
    static constexpr int OFF_WR = IMG_BYTES;                      // float wr_c[2048]
    static constexpr int LDS_BYTES = OFF_WR + RB_M * 4;           // 155648
    static __device__ __forceinline__ int addr(int pos, int cp) { return (pos >> 3) * BLK_BYTES + (pos & 7) * ROW_BYTES + cp * 16; }
    static __device__ __forceinline__ int tw_addr(int e) { return (e >> 4) * BLK_BYTES + 8 * ROW_BYTES + (e & 15) * 8; }   // e < 2048
};

__device__ __forceinline__ void rb_tile_load(const float2 *src /* wave-uniform */, int col_base, const float *wd, float4 (&v)[16],
                                             float2 &wdv, bool valid)
{
    const int w = wave_id();
    int l = threadIdx.x & 63;
    asm volatile("" : "+v"(l));
    const int p0 = w * 8 + (l >> 3), cp = l & 7;
    const rsrc_t rs = make_rsrc(src, valid ? (unsigned)RB_M * RB_N * 8u : 0u);
    const int voff = (p0 * RB_N + col_base + cp * 2) * 8;
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = buf_load_f4<AUX_NT>(rs, voff, 128 * r * RB_N * 8);
    wdv = buf_load_f2<0>(make_rsrc(wd, (unsigned)RB_N * 4u), (col_base + cp * 2) * 4, 0);
}

// The lane's fifteen stage-1 twiddles W_2048^{p0 k1}: k1 = 1 .. 8 come from the table, k1 = 9 .. 15 are the products
// W^{8 p0} W^{(k1 - 8) p0} -- the fused launch (wrp_fused_b.h) has LDS for eight per lane only, and the two forms perform
// the same arithmetic so that their results are bit-identical.  One batch of reads in front of the first butterfly.
__device__ __forceinline__ void rb_stage1_tables(const unsigned char *smem, cf (&tw)[16])
{
    typedef RangeTileB T;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int p0 = (tid >> 6) * 8 + ((tid & 63) >> 3);
#pragma unroll
    for (int k1 = 1; k1 <= 8; k1++) tw[k1] = *reinterpret_cast<const float2 *>(smem + T::tw_addr((p0 * k1) & (RB_M - 1)));
#pragma unroll
    for (int j = 1; j < 8; j++) tw[8 + j] = cmul(tw[8], tw[j]);
}

// stage 1 of ONE of the lane's two columns: window, radix 16, twiddle; k1 < 8 to LDS, k1 >= 8 kept
template <int COLUMN>
__device__ __forceinline__ void rb_stage1(unsigned char *smem, const float4 (&v)[16], float2 wdv, const cf (&tw)[16], cf (&g)[8])
{
    typedef RangeTileB T;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int w = tid >> 6, l = tid & 63, cp = l & 7;
    const int p0 = w * 8 + (l >> 3);
    const float *s_wr = reinterpret_cast<const float *>(smem + T::OFF_WR);
    const int slot = T::addr(p0, cp) + 8 * COLUMN;   // position k1*128 + p0 is 16 k1 blocks further on
    cf a[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {   // (a batch of the sixteen window reads costs this kernel 19 spilled registers)
        const float wgt = s_wr[p0 + 128 * r] * (COLUMN ? wdv.y : wdv.x);
        a[r] = COLUMN ? cscale(make_float2(v[r].z, v[r].w), wgt) : cscale(make_float2(v[r].x, v[r].y), wgt);
    }
    fft16<-1>(a);
    *reinterpret_cast<float2 *>(smem + slot) = a[0];
#pragma unroll
    for (int k1 = 1; k1 < 8; k1++) *reinterpret_cast<float2 *>(smem + slot + k1 * 16 * T::BLK_BYTES) = cmul(a[k1], tw[k1]);
#pragma unroll
    for (int k1 = 8; k1 < 16; k1++) g[k1 - 8] = cmul(a[k1], tw[k1]);
}

__device__ __forceinline__ void rb_group1_to_lds(unsigned char *smem, const cf (&ga)[8], const cf (&gc)[8])
{
    typedef RangeTileB T;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int w = tid >> 6, l = tid & 63, cp = l & 7;
    const int p0 = w * 8 + (l >> 3);
#pragma unroll
    for (int j = 0; j < 8; j++)
        *reinterpret_cast<float4 *>(smem + T::addr(j * 128 + p0, cp)) = make_float4(ga[j].x, ga[j].y, gc[j].x, gc[j].y);
}

// stage 2: one item per lane -- sub-transform kl = w >> 1, p1 = (l >> 4) + 4 (w & 1), column l & 15
__device__ __forceinline__ void rb_stage2(unsigned char *smem)
{
    typedef RangeTileB T;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int w = tid >> 6, l = tid & 63, col = l & 15;
    const int kl = w >> 1, p1 = (l >> 4) + 4 * (w & 1);
    unsigned char *base = smem + kl * 16 * T::BLK_BYTES + p1 * T::ROW_BYTES + col * 8;   // position kl*128 + p1
    cf a[16];
#pragma unroll
    for (int r = 0; r < 16; r++) a[r] = *reinterpret_cast<const float2 *>(base + r * T::BLK_BYTES);
    fft16<-1>(a);
    *reinterpret_cast<float2 *>(base) = a[0];
#pragma unroll
    for (int k2 = 1; k2 < 16; k2++) {   // twiddles at their points of use: the next tile's 64 registers are in flight here
        const cf t = *reinterpret_cast<const float2 *>(smem + T::tw_addr((16 * p1 * k2) & (RB_M - 1)));
        *reinterpret_cast<float2 *>(base + k2 * T::BLK_BYTES) = cmul(a[k2], t);
    }
}

// stage 3 + stores: two items per lane -- k2 = (l >> 4) + 4 (w & 1) + 8 it
__device__ __forceinline__ void rb_stage3_store(unsigned char *smem, float2 *dst /* wave-uniform */, int col_base, int group)
{
    typedef RangeTileB T;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int w = tid >> 6, l = tid & 63, col = l & 15;
    const int kl = w >> 1;
    const rsrc_t rd = make_rsrc(dst, (unsigned)(RB_M / 2) * RB_N * 8u);
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const int k2 = (l >> 4) + 4 * (w & 1) + 8 * it;
        const unsigned char *base = smem + (kl * 16 + k2) * T::BLK_BYTES + col * 8;
        cf a[8];
#pragma unroll
        for (int r = 0; r < 8; r++) a[r] = *reinterpret_cast<const float2 *>(base + r * T::ROW_BYTES);
        fft8<-1>(a);
        const int voff = ((kl + 8 * group + 16 * k2) * RB_N + col_base + col) * 8;
#pragma unroll
        for (int k3 = 0; k3 < 4; k3++) {
            v2f t;
            t.x = a[k3].x; t.y = a[k3].y;
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, t), rd, voff + 256 * k3 * RB_N * 8, 0, AUX_NT);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

__global__ __launch_bounds__(RangeTileB::THREADS, 4) void range_pass_2048(
    const float2 *__restrict__ iq,   // [S][C][2048][128]
    float2 *__restrict__ mid,        // [S][2][1024][128]
    RangeConsts rc, int channels, int total_tiles, const unsigned *gate)
{
    typedef RangeTileB T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (gate_closed(gate)) return;
    constexpr int tiles = RB_N / 16;
    auto decode = [&](int b, int &tile, int &ch, int &sec) {
        tile = b % tiles; b /= tiles;
        ch = b % 2;       b /= 2;
        sec = b;
    };
    float4 v[16];
    float2 wdv;
    int b = blockIdx.x, tile, ch, sec;
    decode(b < total_tiles ? b : 0, tile, ch, sec);
    rb_tile_load(iq + ((size_t)sec * channels + ch) * RB_M * (size_t)RB_N, tile * 16, rc.wd, v, wdv, b < total_tiles);
    for (int e = threadIdx.x; e < RB_M; e += T::THREADS) {
        *reinterpret_cast<float2 *>(smem + T::tw_addr(e)) = rc.tw[e];
        reinterpret_cast<float *>(smem + T::OFF_WR)[e] = rc.wr_c[e];
    }
    __syncthreads();
#pragma unroll 1
    for (; b < total_tiles; b += gridDim.x) {
        float2 *dst = mid + ((size_t)sec * 2 + ch) * (RB_M / 2) * (size_t)RB_N;
        const int col_base = tile * 16;
        cf ga[8], gc[8];
        {
            cf tw[16];
            rb_stage1_tables(smem, tw);
            rb_stage1<0>(smem, v, wdv, tw, ga);
            __builtin_amdgcn_sched_barrier(0);
            rb_stage1<1>(smem, v, wdv, tw, gc);
        }
        __syncthreads();                 // group 0 in the image
        rb_stage2(smem);
        __syncthreads();
        rb_stage3_store(smem, dst, col_base, 0);
        __syncthreads();                 // group 0 has left the image
        rb_group1_to_lds(smem, ga, gc);
        // v, ga, gc are free: request the next tile, it flies during the second half of this one
        const int nb = b + gridDim.x;
        const bool nvalid = nb < total_tiles;
        decode(nvalid ? nb : b, tile, ch, sec);
        rb_tile_load(iq + ((size_t)sec * channels + ch) * RB_M * (size_t)RB_N, tile * 16, rc.wd, v, wdv, nvalid);
        __syncthreads();                 // group 1 in the image
        rb_stage2(smem);
        __syncthreads();
        rb_stage3_store(smem, dst, col_base, 1);
        __syncthreads();                 // image free for the next tile
    }
}

// ================= Doppler rows of 128 pulses: 16 lanes per row, 128 = 8 x 4 x 4 =================
constexpr int DB_WAVES = 4;
// Row buffers: ONE element of padding per FOUR, and 1408 bytes from one row's buffer to the next.  A wave holds four rows; its
// ds_read_b64 are served 32 lanes = two rows at a time over 64 banks, its ds_write_b64 one row at a time over 32: with this
// map the reads of stages 2 and 3 and the writes of stage 2 take their minimum and the stage-1 writes two passes -- 128 LDS
// cycles per pass of four rows where 2 per 16 (round 4) took 160 and the minimum is 96 (tools/lds_banks.py model; round 5:
// SQ_LDS_BANK_CONFLICT was 19 % of the 2048 x 128 launch's LDS-active cycles).
constexpr int DB_ROW_ELEMS = 176;           // complex elements per row buffer: 128 + 31 of padding, rounded so that rows are 1408 bytes apart
__device__ __forceinline__ int db_idx(int pos) { return pos + (pos >> 2); }
__device__ __forceinline__ int db_fidx(int j) { return j + 4 * (j >> 3); }     // as dp_fidx

// sum over the 16 lanes of a DPP row (every lane of the row gets the sum)
__device__ __forceinline__ float row16_sum(float v)
{
    v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);   // row_half_mirror
    v += dpp_mov<0x140>(v);   // row_mirror
    return v;
}

// a4 .. a8 of one row held by 16 lanes (lane i of the row: j = i + 16 r); returns S in every lane of the row.
// buf: this row's LDS buffer; tw: exp(+2 pi i k / 128).
// DUMP: the stage dumps of wrp_dump_stage from THESE kernels (gate, do_dump, dump as in doppler_row)
template <int TAPS, bool DUMP = false>
__device__ __forceinline__ float doppler_row_128(cf (&v)[8], float2 *buf, const float2 *tw, const MaTaps &taps, int i, int gate = 0,
                                                 bool do_dump = false, const DumpPtrs &dump = DumpPtrs{})
{
    float *fbuf = reinterpret_cast<float *>(buf);
    asm volatile("" : "+v"(i));
    float sr = 0.f, si = 0.f;
#pragma unroll
    for (int r = 0; r < 8; r++) { sr += v[r].x; si += v[r].y; }
    sr = row16_sum(sr) * (1.0f / RB_N);
    si = row16_sum(si) * (1.0f / RB_N);
#pragma unroll
    for (int r = 0; r < 8; r++) { v[r].x -= sr; v[r].y -= si; }
    // stage 1: radix 8 over r (j = i + 16 r), twiddle W^{i k1}, to position k1*16 + i
    fft8<+1>(v);
    buf[db_idx(i)] = v[0];
#pragma unroll
    for (int k1 = 1; k1 < 8; k1++) buf[db_idx(k1 * 16 + i)] = cmul(v[k1], tw[(i * k1) & (RB_N - 1)]);
    wave_lds_fence();
    // stage 2: radix 4 over i1 (i = i0 + 4 i1) for (k1, i0); lane takes k1 = (i >> 2) + 4 it, i0 = i & 3
    {
        const int i0 = i & 3;
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const int k1 = (i >> 2) + 4 * it;
            cf x0 = buf[db_idx(k1 * 16 + i0)], x1 = buf[db_idx(k1 * 16 + i0 + 4)], x2 = buf[db_idx(k1 * 16 + i0 + 8)],
               x3 = buf[db_idx(k1 * 16 + i0 + 12)];
            fft4<+1>(x0, x1, x2, x3);
            // twiddle W_16^{i0 k2} = W_128^{8 i0 k2}; result to position k1*16 + i0 + 4 k2 (in place)
            v[4 * it + 0] = x0;
            v[4 * it + 1] = cmul(x1, tw[(8 * i0) & (RB_N - 1)]);
            v[4 * it + 2] = cmul(x2, tw[(16 * i0) & (RB_N - 1)]);
            v[4 * it + 3] = cmul(x3, tw[(24 * i0) & (RB_N - 1)]);
        }
        wave_lds_fence();   // every lane of the row has read before anyone overwrites
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const int k1 = (i >> 2) + 4 * it;
#pragma unroll
            for (int k2 = 0; k2 < 4; k2++) buf[db_idx(k1 * 16 + i0 + 4 * k2)] = v[4 * it + k2];
        }
    }
    wave_lds_fence();
    // stage 3: radix 4 over i0 for (k1, k2); lane takes k1 = (i >> 2) + 4 it, k2 = i & 3; bin k = k1 + 8 k2 + 32 k3
    {
        const int k2 = i & 3;
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const int k1 = (i >> 2) + 4 * it;
            cf x0 = buf[db_idx(k1 * 16 + 4 * k2)], x1 = buf[db_idx(k1 * 16 + 4 * k2 + 1)], x2 = buf[db_idx(k1 * 16 + 4 * k2 + 2)],
               x3 = buf[db_idx(k1 * 16 + 4 * k2 + 3)];
            fft4<+1>(x0, x1, x2, x3);
            v[4 * it + 0] = x0; v[4 * it + 1] = x1; v[4 * it + 2] = x2; v[4 * it + 3] = x3;
        }
    }
    wave_lds_fence();   // everyone has read before the buffer is reused for |.|^2
    // shift (swap halves), clip post-shift bins n-1, n-2, |.|^2 in natural order
    float part = 0.f;
    {
        const int k2 = i & 3;
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const int k1 = (i >> 2) + 4 * it;
#pragma unroll
            for (int k3 = 0; k3 < 4; k3++) {
                const int k = k1 + 8 * k2 + 32 * k3;
                const int j = (k + RB_N / 2) & (RB_N - 1);
                cf z = v[4 * it + k3];
                if (DUMP && do_dump && dump.noshift) dump.noshift[(size_t)gate * RB_N + k] = make_float2(z.x, -z.y);   // before the final conj
                if (j >= RB_N - 2) z = make_float2(0.f, 0.f);
                if (DUMP && do_dump && dump.fft2) dump.fft2[(size_t)gate * RB_N + j] = z;
                const float p2 = fmaf(z.y, z.y, z.x * z.x);
                part += p2;
                if (DUMP) fbuf[db_fidx(j)] = p2;
            }
        }
    }
    // a7 + a8: the row sum of the circular moving average is (sum of the taps) x (sum of |.|^2) -- see doppler_row
    // (wrp_kernels.h); the convolution itself is formed by the DUMP instantiation only, for the 08pow stage
    const float S = row16_sum(part) * taps.sum;
    if (DUMP) {
        wave_lds_fence();
        // a7 as a stage: lane owns bins 8 i .. 8 i + 7 plus an 8-bin halo (circular over the 16 lanes of the row)
        float a[16];
        {
            const float4 *f4 = reinterpret_cast<const float4 *>(fbuf);
            const int prev = 3 * ((i + 15) & 15);
            const float4 h0 = f4[prev], h1 = f4[prev + 1], c0 = f4[3 * i], c1 = f4[3 * i + 1];
            a[0] = h0.x; a[1] = h0.y; a[2] = h0.z; a[3] = h0.w;
            a[4] = h1.x; a[5] = h1.y; a[6] = h1.z; a[7] = h1.w;
            a[8] = c0.x; a[9] = c0.y; a[10] = c0.z; a[11] = c0.w;
            a[12] = c1.x; a[13] = c1.y; a[14] = c1.z; a[15] = c1.w;
        }
        if (do_dump && dump.abs2) {
#pragma unroll
            for (int u = 0; u < 8; u++) dump.abs2[(size_t)gate * RB_N + 8 * i + u] = a[8 + u];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            float p = 0.f;
#pragma unroll
            for (int t = 0; t < TAPS; t++) p = fmaf(taps.g[t], a[8 + u - t], p);
            if (do_dump && dump.pow) dump.pow[(size_t)gate * RB_N + 8 * i + u] = p;
        }
        if (do_dump && dump.rowsum && i == 0) dump.rowsum[gate] = S;
        wave_lds_fence();
    }
    return S;
}

// gridDim.y = n_sectors, or a few rows that walk the sectors (the gated repeat of a batch: see doppler_pass_512)
template <int TAPS, bool DUMP = false>
__global__ __launch_bounds__(DB_WAVES * 64) void doppler_pass_128(
    const float2 *__restrict__ mid,  // [S][2][gates][128]
    float *__restrict__ out,         // [S][gates][2]
    const float2 *__restrict__ tw,   // [128] exp(+2 pi i k / 128)
    int gates, int n_sectors, MaTaps taps, float k_rr, float k_cal, DumpPtrs dump, const unsigned *gate_word)
{
    __shared__ __attribute__((aligned(16))) float2 lds[DB_WAVES * 4][DB_ROW_ELEMS];
    __shared__ __attribute__((aligned(16))) float2 s_tw[RB_N];
    if (gate_closed(gate_word)) return;
    const int w = wave_id(), l = threadIdx.x & 63;
    const int sub = l >> 4, i = l & 15;               // sub: 0 (g, HH), 1 (g, VV), 2 (g + 1, HH), 3 (g + 1, VV)
    const int gate = (blockIdx.x * DB_WAVES + w) * 2 + (sub >> 1), ch = sub & 1;
    bool tables = false;
#pragma unroll 1
    for (int sec = blockIdx.y; sec < n_sectors; sec += gridDim.y) {
        // descriptor on the sector (wave-uniform); the row of this 16-lane group goes into the lane offset
        const rsrc_t rs = make_rsrc(mid + (size_t)sec * 2 * gates * RB_N, (unsigned)(2 * gates * RB_N) * 8u);
        const int voff = ((ch * gates + gate) * RB_N + i) * 8;
        cf x[8];
#pragma unroll
        for (int r = 0; r < 8; r++) x[r] = buf_load_f2<AUX_NT>(rs, voff, 16 * r * 8);
        if (!tables) {                   // (workgroup-uniform)
            for (int e = threadIdx.x; e < RB_N; e += DB_WAVES * 64) s_tw[e] = tw[e];
            __syncthreads();
            tables = true;
        }
        const float S = doppler_row_128<TAPS, DUMP>(x, lds[w * 4 + sub], s_tw, taps, i, gate, DUMP && dump.channel == ch && sec == 0, dump);
        const float other = __shfl(S, (l + 16) & 63);     // the VV row sum sits 16 lanes above the HH one
        if (i == 0 && ch == 0) {
            unsigned hdr;
            unsigned *frames = sector_frames(dump, sec, gates, hdr);
            reflectivity_store(&out[((size_t)sec * gates + gate) * 2], gate, S, other, k_rr, k_cal, frames, gates, hdr);
        }
    }
}

} // namespace wrp
