// wrp_kernels.h -- the HIP kernels of the per-sector chain (gfx950, wave64).
//
//   range_pass_1024   : Hamming window (a2) + range FFT along i (a3), one workgroup per
//                       (sector, channel, column tile); writes gates k < m/2 only.
//   doppler_pass_512  : mean removal (a4), Doppler FFT + conj + shift + clip (a5), |.|^2 (a6),
//                       7-tap causal circular MA (a7), row sum (a8), Zdb/Zdr (a9); one wave per
//                       range gate, both polarisations in the same wave.
//   (wrp_fused.h / wrp_fused_b.h: the same device functions inside ONE persistent launch whose XCD teams keep the
//    half-height intermediate of a sector in that XCD's L2 -- the default for batches of >= 8 sectors.)
//
// Reference semantics: read.cc:133-345 / rpv2.cu:86-213,409-570 (DESIGN.md §1 maps every stage).
// No rocFFT/hipFFT: the FFTs are LDS-resident mixed-radix passes (16x8x8 for m = 1024, 8x8x8 for
// n = 512) built from fft_radix.h.  All forms of a pass share the device functions below, factor the
// transforms the same way and the library is built with -ffp-contract=off, so they are bit-identical to
// each other (tests/test_gpu_parity.py).
#pragma once
#include <hip/hip_runtime.h>

#include "fft_radix.h"

namespace wrp {

struct DumpPtrs {       // all optional (nullptr = skip); one sector, one channel
    float2 *hamm;       // [m][n]
    float2 *fft1;       // [m][n]
    float2 *noshift;    // [m/2][n]
    float2 *fft2;       // [m/2][n]
    float *abs2;        // [m/2][n]
    float *pow;         // [m/2][n]
    float *rowsum;      // [m/2]
    int channel;        // which channel the dump refers to
    // not a dump: the products framed for the wire -- per sector two planes of 1 + m/2 words,
    // [header][m/2 BIG-ENDIAN floats], Zdb then Zdr (rpv2.cu:631-661 does the swap on the CPU, aftoab); nullptr = none.
    // frame_hdrs == nullptr: sector 0 of the launch only, header frame_hdr (the slot path); otherwise every sector s of the
    // launch, frames + s * 2 * (1 + m/2), header frame_hdrs[s] (the batch entries)
    unsigned *frames;
    unsigned frame_hdr; // the header word as it lies in memory: sector BE16, elevation BE16
    const unsigned *frame_hdrs;
};
// where the frames of sector `sec` of the launch go (nullptr: nowhere) and their header word
__device__ __forceinline__ unsigned *sector_frames(const DumpPtrs &d, int sec, int gates, unsigned &hdr)
{
    hdr = d.frame_hdr;
    if (!d.frames) return nullptr;
    if (!d.frame_hdrs) return sec == 0 ? d.frames : nullptr;
    hdr = d.frame_hdrs[sec];
    return d.frames + (size_t)sec * 2 * (1 + gates);
}
// a batch launch that only runs when the fused launch in front of it on the stream has given up: *gate is that launch's
// status word in device memory (0 = it succeeded, there is nothing to repeat); nullptr = not gated
__device__ __forceinline__ bool gate_closed(const unsigned *gate /* wave-uniform */) { return gate && *gate == 0; }

struct MaTaps { float g[9]; float sum; };   // sum = g[0] + ... + g[count-1], formed in double from the rounded taps

struct RangeConsts {
    const float *wr_c;   // [1024]  range window * c
    const float *wd;     // [n]     Doppler window
    const float2 *tw;    // [1024]  exp(-2 pi i k / 1024)
};

// ---- buffer addressing: descriptor (SGPRs) + ONE 32-bit lane offset + scalar offset ---------
// The compiler turns `uniform_ptr + const + lane` into per-access 64-bit VGPR addresses (32 VGPRs
// for a 16-load tile); raw buffer operations keep base and per-access offset in SGPRs.  The
// descriptor must be built from values the compiler KNOWS are wave-uniform (readfirstlane), or
// every access becomes a waterfall loop.
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
// Cache-policy bits of the raw buffer intrinsics on gfx94x/gfx950: 1 = sc0, 2 = nt, 16 = sc1.
// Measured on MI355X (tools/l2handoff.hip): an sc0 load still HITS the CU's L1 (workgroup scope; a
// poll on a counter written by another CU never saw the update); an sc1 load (device scope) misses
// the L1 and is served by the XCD's L2 at full speed (1.5 TB/s per XCD) -- the right load for a
// hand-off between CUs of one XCD; `buffer_inv sc1` + plain loads works too but the invalidate
// costs 2.4x the time; nt marks a line as the first to be evicted.
constexpr int AUX_SC0 = 1;
constexpr int AUX_NT = 2;    // non-temporal: the line is the first to leave the caches again
constexpr int AUX_SC1 = 16;

__device__ __forceinline__ rsrc_t make_rsrc(const void *p, unsigned bytes)   // p, bytes wave-uniform
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
// NB: convert whole vectors -- element-wise __builtin_bit_cast(float, u.x) on the builtin's result
// makes hipcc (ROCm 7.2) shrink the access to a single dword.
template <int AUX = 0>
__device__ __forceinline__ float4 buf_load_f4(rsrc_t r, int voff, int soff)
{
    const v4f f = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX));
    return make_float4(f.x, f.y, f.z, f.w);
}
// ALWAYS pass soff = 0 (an immediate): with an SGPR soffset hipcc (ROCm 7.2) schedules a VALU
// write to the first data VGPR directly behind the 128-bit store without the wait state gfx9
// requires, and the store then carries the new value (seen as 0.07 dB errors in isolated gates).
template <int AUX = 0>
__device__ __forceinline__ void buf_store_f4(rsrc_t r, int voff, int soff, float4 f)
{
    v4f t;
    t.x = f.x; t.y = f.y; t.z = f.z; t.w = f.w;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, t), r, voff, soff, AUX);
}
// Cache policy of the two-kernel path's streams (input tiles in, intermediate out, intermediate
// in): every byte is touched once per kernel, so they are non-temporal -- measured 3.53 vs 3.67
// us/sector (A/B in one session, twice).  Exception: 8-column tiles cover half a 128-byte line and
// rely on the neighbouring tile finding the other half in the L2 (nt: 4.7 vs 2.8 us/sector).
template <int TCOLS> struct StreamAux { static constexpr int value = TCOLS == 16 ? 2 /* AUX_NT */ : 0; };
template <int AUX>
__device__ __forceinline__ float2 buf_load_f2(rsrc_t r, int voff, int soff)
{
    const v2f f = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, AUX));
    return make_float2(f.x, f.y);
}
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// =============================================================================================
// range tile, m = 1024 = 16 x 8 x 8; TCOLS = 16 columns per 512-thread workgroup or 8 columns
// per 256-thread workgroup.
//
// In-place decimation-in-frequency over positions p of a column:
//   stage 1 (registers, straight from HBM): lane owns rows p0 + 64 r, r < 16   -> radix 16,
//           twiddle W_1024^{p0 k1}, result to LDS position k1*64 + p0
//   stage 2 (LDS): rows k1*64 + p1 + 8 r, r < 8  -> radix 8, twiddle W_64^{p1 k2}, in place
//   stage 3 (LDS): rows k1*64 + k2*8 + r, r < 8  -> radix 8; output row k = k1 + 16 k2 + 128 k3
// LDS image: [position][TCOLS columns] complex, plus one position of padding after every 8
// positions so that stage 3's ds_read_b128 (lanes = column pairs x 8 k2) is bank-conflict free;
// stages 1 and 2 touch whole contiguous 1 KiB rows per wave-instruction.  The padding is not
// wasted: the 1024-entry twiddle table lives in it (tw_addr), so twiddles are LDS reads
// (lgkmcnt) instead of dependent global loads (vmcnt, in order behind the tile prefetch).
// The range window wr_c (4 KiB) sits right behind the image.
// =============================================================================================
constexpr int RP_M = 1024;

template <int TCOLS>
struct RangeTile {
    static constexpr int CP = TCOLS / 2;              // column pairs (one float4) per row segment
    static constexpr int ROWS_PER_WAVE = 64 / CP;     // row segments one wave-instruction covers
    static constexpr int WAVES = 64 / ROWS_PER_WAVE;  // waves so that the block covers 64 rows
    static constexpr int THREADS = 64 * WAVES;        // 512 (16 columns) or 256 (8 columns)
    static constexpr int ROW_BYTES = TCOLS * 8;       // bytes of one position in LDS
    static constexpr int BLK_BYTES = 9 * ROW_BYTES;   // 8 positions + one position of padding
    static constexpr int IMG_BYTES = (RP_M / 8) * BLK_BYTES;   // 147456 / 73728
    static constexpr int OFF_WR = IMG_BYTES;                    // float wr_c[1024]
    static constexpr int LDS_BYTES = IMG_BYTES + RP_M * 4;      // 151552 / 77824
    static constexpr int TW_PER_PAD = ROW_BYTES / 8;            // twiddle entries per padding row (16 / 8)
    static constexpr int TW_BLK0 = TCOLS == 16 ? 64 : 0;        // 16 cols: pads of blocks 64..127 (blocks 0..63
                                                                // are re-used as wave buffers by the fused launch)
    static __device__ __forceinline__ int addr(int pos, int colpair)   // byte address of a float4
    {
        return (pos >> 3) * BLK_BYTES + (pos & 7) * ROW_BYTES + colpair * 16;
    }
    static __device__ __forceinline__ int tw_addr(int e)               // byte address of twiddle e < 1024
    {
        return (TW_BLK0 + e / TW_PER_PAD) * BLK_BYTES + 8 * ROW_BYTES + (e % TW_PER_PAD) * 8;
    }
};

// copy the twiddle and window tables into LDS (once per workgroup; caller barriers afterwards)
template <int TCOLS>
__device__ __forceinline__ void range_tables_to_lds(unsigned char *smem, const RangeConsts &rc)
{
    typedef RangeTile<TCOLS> T;
    for (int e = threadIdx.x; e < RP_M; e += T::THREADS) {
        *reinterpret_cast<float2 *>(smem + T::tw_addr(e)) = rc.tw[e];
        reinterpret_cast<float *>(smem + T::OFF_WR)[e] = rc.wr_c[e];
    }
}

// issue the 16 row loads of this lane's two columns (rows p0 + 64 r) and its two Doppler-window values
template <int TCOLS>
// valid = false (wave-uniform): the descriptor gets zero records, the hardware drops all 17 loads
// (they return 0) -- a branch-free way to skip a prefetch, because hipcc waits vmcnt(0) for
// everything in flight at the first use behind a load that sits in a conditional block.
__device__ __forceinline__ void range_load(const float2 *src /* wave-uniform */, int n, int col_base,
                                           const float *wd, float4 (&v)[16], float2 &wdv, bool valid = true)
{
    typedef RangeTile<TCOLS> T;
    const int w = wave_id(), l = threadIdx.x & 63;
    const int p0 = w * T::ROWS_PER_WAVE + l / T::CP;
    const rsrc_t rs = make_rsrc(src, valid ? (unsigned)RP_M * n * 8u : 0u);
    const int voff = (p0 * n + col_base + (l % T::CP) * 2) * 8;
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = buf_load_f4<StreamAux<TCOLS>::value>(rs, voff, 64 * r * n * 8);
    wdv = buf_load_f2<0>(make_rsrc(wd, (unsigned)n * 4u), (col_base + (l % T::CP) * 2) * 4, 0);
}

struct NoHook { __device__ __forceinline__ void operator()() const {} };

// stage 1 + stage 2 (ends with the tile in LDS, positions k1*64 + p1 + 8 k2, after a barrier).
// after_stage1() runs once v has been consumed (behind the first barrier): the persistent
// kernels use it to request the NEXT tile into v while stages 2 and 3 of this one run.
template <int TCOLS, bool DUMP, class Hook = NoHook>
__device__ __forceinline__ void range_stage12(unsigned char *smem, float4 (&v)[16], float2 wdv, int n, int col_base,
                                              bool do_dump, const DumpPtrs &dump, Hook after_stage1 = Hook())
{
    typedef RangeTile<TCOLS> T;
    const int tid = threadIdx.x;
    {
        const int w = tid >> 6, l = tid & 63;
        const int cp = l % T::CP;
        int p0 = w * T::ROWS_PER_WAVE + l / T::CP;
        // opaque to the optimiser: inside the persistent launch the ~50 per-lane LDS addresses
        // derived from p0 / p1 / k2 would otherwise be hoisted out of the task loop and spilled
        asm volatile("" : "+v"(p0));
        const int col0 = col_base + cp * 2;
        const float *s_wr = reinterpret_cast<const float *>(smem + T::OFF_WR);
        cf a[16], c[16];
        float wrow[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            wrow[r] = s_wr[p0 + 64 * r];
            a[r] = make_float2(v[r].x, v[r].y);
            c[r] = make_float2(v[r].z, v[r].w);
        }
        if (DUMP && do_dump && dump.hamm) {   // a2 as a stage of its own: the production form folds it into the butterflies below
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const cf ha = cscale(a[r], wrow[r] * wdv.x), hc = cscale(c[r], wrow[r] * wdv.y);
                *reinterpret_cast<float4 *>(&dump.hamm[(size_t)(p0 + 64 * r) * n + col0]) = make_float4(ha.x, ha.y, hc.x, hc.y);
            }
        }
        cf tw1[16];   // all 15 twiddles requested in one batch, ahead of the butterflies (one LDS latency)
#pragma unroll
        for (int k1 = 1; k1 < 16; k1++)
            tw1[k1] = *reinterpret_cast<const float2 *>(smem + T::tw_addr((p0 * k1) & (RP_M - 1)));
        fft16_scaled<-1>(a, wrow, wdv.x);
        fft16_scaled<-1>(c, wrow, wdv.y);
        *reinterpret_cast<float4 *>(smem + T::addr(p0, cp)) = make_float4(a[0].x, a[0].y, c[0].x, c[0].y);
#pragma unroll
        for (int k1 = 1; k1 < 16; k1++) {
            const cf t = tw1[k1];
            const cf x = cmul(a[k1], t), y = cmul(c[k1], t);
            *reinterpret_cast<float4 *>(smem + T::addr(k1 * 64 + p0, cp)) = make_float4(x.x, x.y, y.x, y.y);
        }
    }
    __syncthreads();
    after_stage1();
    const int cp = tid % T::CP, kb = tid / (T::CP * 8);
    int p1 = (tid / T::CP) & 7;
    asm volatile("" : "+v"(p1));
    cf tw2[8];   // W_64^{p1 k2}: one batch of LDS reads for both items
#pragma unroll
    for (int k2 = 1; k2 < 8; k2++)
        tw2[k2] = *reinterpret_cast<const float2 *>(smem + T::tw_addr((16 * p1 * k2) & (RP_M - 1)));
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const int k1 = kb + 8 * it;
        cf a[8], c[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const float4 u = *reinterpret_cast<const float4 *>(smem + T::addr(k1 * 64 + p1 + 8 * r, cp));
            a[r] = make_float2(u.x, u.y);
            c[r] = make_float2(u.z, u.w);
        }
        fft8<-1>(a);
        fft8<-1>(c);
        *reinterpret_cast<float4 *>(smem + T::addr(k1 * 64 + p1, cp)) = make_float4(a[0].x, a[0].y, c[0].x, c[0].y);
#pragma unroll
        for (int k2 = 1; k2 < 8; k2++) {
            const cf t = tw2[k2];
            const cf x = cmul(a[k2], t), y = cmul(c[k2], t);
            *reinterpret_cast<float4 *>(smem + T::addr(k1 * 64 + p1 + 8 * k2, cp)) = make_float4(x.x, x.y, y.x, y.y);
        }
    }
    __syncthreads();
}

// stage 3: last radix-8 and the store of gates k < m/2 (the chain never reads the rest, rpv2.cu:502)
template <int TCOLS, bool DUMP>
__device__ __forceinline__ void range_stage3(const unsigned char *smem, float2 *dst /* wave-uniform */, int n,
                                             int col_base, bool do_dump, const DumpPtrs &dump)
{
    typedef RangeTile<TCOLS> T;
    const int tid = threadIdx.x;
    const int cp = tid % T::CP, kb = tid / (T::CP * 8);
    int k2 = (tid / T::CP) & 7;
    asm volatile("" : "+v"(k2));
    const int col0 = col_base + cp * 2;
    const rsrc_t rd = make_rsrc(dst, (unsigned)(RP_M / 2) * n * 8u);
    const int voff = ((kb + 16 * k2) * n + col0) * 8;   // row k1 + 16 k2 at it = 0; scalar offsets below
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const int k1 = kb + 8 * it;
        cf a[8], c[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const float4 u = *reinterpret_cast<const float4 *>(smem + T::addr(k1 * 64 + k2 * 8 + r, cp));
            a[r] = make_float2(u.x, u.y);
            c[r] = make_float2(u.z, u.w);
        }
        fft8<-1>(a);
        fft8<-1>(c);
        const int k0 = k1 + 16 * k2;
#pragma unroll
        for (int k3 = 0; k3 < 4; k3++)   // row offset in the VGPR, soffset 0: see buf_store_f4
            buf_store_f4<StreamAux<TCOLS>::value>(rd, voff + (8 * it + 128 * k3) * n * 8, 0, make_float4(a[k3].x, a[k3].y, c[k3].x, c[k3].y));
        if (DUMP && do_dump && dump.fft1) {
#pragma unroll
            for (int k3 = 0; k3 < 8; k3++)
                *reinterpret_cast<float4 *>(&dump.fft1[(size_t)(k0 + 128 * k3) * n + col0]) =
                    make_float4(a[k3].x, a[k3].y, c[k3].x, c[k3].y);
        }
    }
}

template <int TCOLS, bool DUMP>
__global__ __launch_bounds__(RangeTile<TCOLS>::THREADS) void range_pass_1024(
    const float2 *__restrict__ iq,   // [S][C][1024][n]
    float2 *__restrict__ mid,        // [S][2][512][n]
    RangeConsts rc, int n, int channels, DumpPtrs dump, const unsigned *gate)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (gate_closed(gate)) return;
    const int tiles = n / TCOLS;
    int b = blockIdx.x;
    if (TCOLS == 8) {
        // two 8-column tiles share every 128-byte line: give the pair to blocks x and x + 8,
        // which the dispatcher places on the same XCD, so that its L2 serves the second read
        // (speed only -- any placement computes the same result)
        const int x = b & 15;
        b = (b & ~15) + ((x & 7) << 1) + (x >> 3);
    }
    const int tile = b % tiles; b /= tiles;
    const int ch = b % 2;       b /= 2;
    const int sec = b;
    const float2 *src = iq + ((size_t)sec * channels + ch) * RP_M * (size_t)n;
    float2 *dst = mid + ((size_t)sec * 2 + ch) * (RP_M / 2) * (size_t)n;
    const bool do_dump = DUMP && dump.channel == ch && sec == 0;
    float4 v[16];
    float2 wdv;
    range_load<TCOLS>(src, n, tile * TCOLS, rc.wd, v, wdv);      // HBM requests first ...
    range_tables_to_lds<TCOLS>(smem, rc);                        // ... tables while they fly
    __syncthreads();
    range_stage12<TCOLS, DUMP>(smem, v, wdv, n, tile * TCOLS, do_dump, dump);
    range_stage3<TCOLS, DUMP>(smem, dst, n, tile * TCOLS, do_dump, dump);
}

// Persistent form of the same pass: a fixed grid (a multiple of 16 blocks, so the XCD pairing of
// the 8-column tiles survives) walks the tiles; the NEXT tile of a workgroup is requested as soon
// as stage 1 has consumed the registers of the current one and flies during stages 2 and 3, so a
// CU always has HBM requests outstanding (the one-tile-per-workgroup form alternates between
// waiting for its loads and computing).  Same device functions -> bit-identical results.
template <int TCOLS>
__global__ __launch_bounds__(RangeTile<TCOLS>::THREADS) void range_pass_1024_persistent(
    const float2 *__restrict__ iq, float2 *__restrict__ mid, RangeConsts rc, int n, int channels, int total_tiles,
    const unsigned *gate)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (gate_closed(gate)) return;
    const int tiles = n / TCOLS;
    const DumpPtrs nodump{};
    auto decode = [&](int b, int &tile, int &ch, int &sec) {
        if (TCOLS == 8) {
            const int x = b & 15;
            b = (b & ~15) + ((x & 7) << 1) + (x >> 3);
        }
        tile = b % tiles; b /= tiles;
        ch = b % 2;       b /= 2;
        sec = b;
    };
    float4 v[16];
    float2 wdv;
    int b = blockIdx.x, tile, ch, sec;
    decode(b < total_tiles ? b : 0, tile, ch, sec);
    range_load<TCOLS>(iq + ((size_t)sec * channels + ch) * RP_M * (size_t)n, n, tile * TCOLS, rc.wd, v, wdv, b < total_tiles);
    range_tables_to_lds<TCOLS>(smem, rc);
    __syncthreads();
#pragma unroll 1
    for (; b < total_tiles; b += gridDim.x) {
        float2 *dst = mid + ((size_t)sec * 2 + ch) * (RP_M / 2) * (size_t)n;
        const int col_base = tile * TCOLS;
        const float2 wcur = wdv;
        const int nb = b + gridDim.x;
        const bool nvalid = nb < total_tiles;
        int ntile, nch, nsec;
        decode(nvalid ? nb : b, ntile, nch, nsec);
        range_stage12<TCOLS, false>(smem, v, wcur, n, col_base, false, nodump, [&]() {
            // branch-free: a zero-record descriptor drops the loads behind the last tile
            range_load<TCOLS>(iq + ((size_t)nsec * channels + nch) * RP_M * (size_t)n, n, ntile * TCOLS, rc.wd, v, wdv, nvalid);
        });
        range_stage3<TCOLS, false>(smem, dst, n, col_base, false, nodump);
        __syncthreads();   // stage 3 has read the image before the next tile's stage 1 overwrites it
        tile = ntile; ch = nch; sec = nsec;
    }
}

// =============================================================================================
// Doppler row, n = 512 = 8 x 8 x 8, one wave per row, wave-private LDS, no workgroup barrier.
// Twiddles come from a 512-entry LDS table at the point of use (14 ds_read_b64 per row).
// =============================================================================================
constexpr int DP_N = 512;
constexpr int DP_WAVES = 4;
constexpr int DP_ELEMS = DP_N + DP_N / 8;   // padded complex elements per wave buffer (576 = 4608 B)

// element index of position p: two elements of padding per 16 (tools/lds_banks.py: stage 1's and stage 2's accesses
// conflict free, stage 3's reads two passes: 112 LDS cycles per row for the three exchanges.  With neighbours swapped where
// bit 3 of the position is set all three are conflict free (96 cycles) and the launch is no faster:
// profiles/r03/ab_doppler_lds_swizzle.log; not kept)
__device__ __forceinline__ int dp_idx(int pos) { return pos + 2 * (pos >> 4); }
// float index of |.|^2 bin j: 4 floats of padding per 8 -> b32 writes and b128 reads conflict free
__device__ __forceinline__ int dp_fidx(int j) { return j + 4 * (j >> 3); }

struct TagTrue { static constexpr bool value = true; };
struct TagFalse { static constexpr bool value = false; };

__device__ __forceinline__ void wave_lds_fence()
{
    // LDS operations of one wave execute in issue order, so lanes may read what other lanes of the
    // same wave wrote earlier without any wait; this only stops the COMPILER from moving memory
    // accesses across the exchange point (no instruction is emitted).
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// sum over the 64 lanes without touching LDS: DPP inside each row of 16, v_readlane across rows
__device__ __forceinline__ float wave_sum(float v)
{
    v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);   // row_half_mirror
    v += dpp_mov<0x140>(v);   // row_mirror
    const int b = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16)) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) +
            __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48)));
}

// lane l takes j = l + 64 r.  SC1 = L1-bypassing loads (rows written by other CUs in this launch)
template <int AUX>   // cache policy bits: 0 plain, AUX_SC0 = miss the CU's L1 and be served by the XCD's L2
__device__ __forceinline__ void doppler_load_row(const float2 *row /* wave-uniform */, int l, cf (&x)[8])
{
    const rsrc_t rs = make_rsrc(row, DP_N * 8u);
#pragma unroll
    for (int r = 0; r < 8; r++) x[r] = buf_load_f2<AUX>(rs, l * 8, 64 * r * 8);
}

// The Doppler twiddles in LDS, ARRANGED per lane so that every read is `lane base + immediate offset` instead of
// an index computation per twiddle, and lane-contiguous so that a wave's b64 read has no bank conflict:
// [8 k][64 lanes] W^{l k} (stage 1), then [8 k][8 p] W^{8 p k} (stage 2), W = exp(+2 pi i / 512).
// The engine keeps the table in this arrangement in device memory (wrp_engine.hip: arranged_doppler_twiddles), so
// filling LDS is a plain coalesced copy.
constexpr int DP_TW_ELEMS = 64 * 8 + 8 * 8;
__host__ __device__ inline int doppler_twiddle_index(int e)   // arranged entry e -> index into exp(+2 pi i k / 512)
{
    if (e < 512) return ((e & 63) * (e >> 6)) & (DP_N - 1);    // stage 1: k = e >> 6, lane = e & 63
    e -= 512;
    return (8 * (e & 7) * (e >> 3)) & (DP_N - 1);              // stage 2: k = e >> 3, p = e & 7
}
__device__ __forceinline__ void doppler_twiddles_to_lds(float2 *s_tw, const float2 *tw_arranged /* [DP_TW_ELEMS] global */, int tid,
                                                        int threads)
{
    for (int e = tid; e < DP_TW_ELEMS; e += threads) s_tw[e] = tw_arranged[e];
}

// a4..a8 for one row held in v (lane l: j = l + 64 r); returns S (the same value in every lane).
// tw: the arranged LDS table of doppler_twiddles_to_lds.
// The lane's fourteen twiddles of a row (seven per stage): the same for every row a wave transforms.  The two-kernel
// pass reads them per row (a wave transforms two); the row waves of the fused launch keep them in registers for their 180
// rows -- 14 of a row's 47 LDS instructions and 7 of its 23 KiB of LDS traffic (the row role has the registers: the
// kernel's allocation is the tile role's).
struct DopplerTwiddles { cf t1[8], t2[8]; };
__device__ __forceinline__ void doppler_row_twiddles(const float2 *tw, int l, DopplerTwiddles &t)
{
#pragma unroll
    for (int k = 1; k < 8; k++) {
        t.t1[k] = tw[k * 64 + l];
        t.t2[k] = tw[512 + k * 8 + (l & 7)];
    }
}
template <bool DUMP, int TAPS, bool KEPT = false>
__device__ __forceinline__ float doppler_row(cf (&v)[8], float2 *buf, const float2 *tw, const MaTaps &taps, int l,
                                             int gate, bool do_dump, const DumpPtrs &dump, const DopplerTwiddles &kept = DopplerTwiddles{})
{
    float *fbuf = reinterpret_cast<float *>(buf);
    asm volatile("" : "+v"(l));   // keep the per-lane LDS addresses inside the row (see range_stage12)
    l &= 63;                      // ... but let the compiler know its range again: (c*64 + l) >> 4 folds only then
    // a4: mean over the row, subtract (rpv2.cu:434-439)
    float sr = 0.f, si = 0.f;
#pragma unroll
    for (int r = 0; r < 8; r++) { sr += v[r].x; si += v[r].y; }
    sr = wave_sum(sr) * (1.0f / DP_N);
    si = wave_sum(si) * (1.0f / DP_N);
#pragma unroll
    for (int r = 0; r < 8; r++) { v[r].x -= sr; v[r].y -= si; }

    // a5: Z[k] = sum_j (x_j - mu) exp(+2 pi i j k / n)   (= conj . FFT . conj)
    cf t1[8], t2[8];                               // both twiddle sets in one batch of LDS reads, or the kept ones
#pragma unroll
    for (int k = 1; k < 8; k++) {
        t1[k] = KEPT ? kept.t1[k] : tw[k * 64 + l];
        t2[k] = KEPT ? kept.t2[k] : tw[512 + k * 8 + (l & 7)];
    }
    // LDS positions below are written as `lane base + compile-time offset`: with l < 64, p1, k2 < 8 the padded index
    // dp_idx(pos) = pos + 2 (pos >> 4) is linear in the unrolled counter (one address register per stage, immediates
    // for the eight elements), which the compiler does not find by itself:
    //   dp_idx(k1 64 + l)            = dp_idx(l) + 72 k1
    //   dp_idx(k1 64 + p1 + 8 r)     = 72 k1 + p1 + 8 r + 2 (r >> 1)
    //   dp_idx(k1 64 + 8 k2 + r)     = 72 k1 + 8 k2 + 2 (k2 >> 1) + r
    fft8<+1>(v);                                   // stage 1: lane l owns j = l + 64 r
    const int b1 = dp_idx(l);
    buf[b1] = v[0];
#pragma unroll
    for (int k1 = 1; k1 < 8; k1++) buf[b1 + 72 * k1] = cmul(v[k1], t1[k1]);
    wave_lds_fence();
    {                                              // stage 2: lane = p1 + 8 k1, positions k1*64 + p1 + 8 r
        const int b2 = 72 * (l >> 3) + (l & 7);
#pragma unroll
        for (int r = 0; r < 8; r++) v[r] = buf[b2 + 8 * r + 2 * (r >> 1)];
        fft8<+1>(v);
        buf[b2] = v[0];
#pragma unroll
        for (int k2 = 1; k2 < 8; k2++) buf[b2 + 8 * k2 + 2 * (k2 >> 1)] = cmul(v[k2], t2[k2]);
    }
    wave_lds_fence();
    // stage 3: lane = k2 + 8 k1 owns positions k1*64 + k2*8 + r; output k = k1 + 8 k2 + 64 k3
    const int k2 = l & 7, k1 = l >> 3;
    const int klo = k1 + 8 * k2;
    const int b3 = 72 * k1 + 8 * k2 + 2 * (k2 >> 1);
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = buf[b3 + r];
    fft8<+1>(v);
    wave_lds_fence();   // everyone has read before the buffer is reused for |.|^2

    if (DUMP && do_dump && dump.noshift) {   // reference dumps the FFT output before the final conj
#pragma unroll
        for (int k3 = 0; k3 < 8; k3++)
            dump.noshift[(size_t)gate * DP_N + klo + 64 * k3] = make_float2(v[k3].x, -v[k3].y);
    }
    // shift (swap halves: j = k + n/2 mod n), clip post-shift bins n-1, n-2, |.|^2
    // a7 + a8: the chain needs the moving average (read.cc:290-301: a CIRCULAR convolution, FFT x H x inverse FFT) only
    // through its row sum (read.cc:303-311), and the sum of a circular convolution is its DC bin:
    //   sum_j sum_t g[t] A[(j - t) mod n] = (sum_t g[t]) (sum_j A[j]).
    // So the launch sums |.|^2 where the Doppler transform leaves it, in registers, and multiplies by the taps' sum; the
    // convolution itself (08pow) is formed by the DUMP instantiations only, for the stage dump -- 56 of a row's 456
    // vector instructions and 12 of its 47 LDS instructions less.  Every launch form shares this function.
    const int fb = klo + 4 * (klo >> 3);
    float part = 0.f;
#pragma unroll
    for (int k3 = 0; k3 < 8; k3++) {
        const int j = ((k3 + 4) & 7) * 64 + klo;
        cf z = v[k3];
        if (j >= DP_N - 2) z = make_float2(0.f, 0.f);
        if (DUMP && do_dump && dump.fft2) dump.fft2[(size_t)gate * DP_N + j] = z;
        const float p2 = fmaf(z.y, z.y, z.x * z.x);
        part += p2;
        if (DUMP) fbuf[fb + 96 * ((k3 + 4) & 7)] = p2;   // = dp_fidx(j): j + 4 (j >> 3) with klo < 64
    }
    const float S = wave_sum(part) * taps.sum;     // a8: row sum
    if (DUMP) {
        wave_lds_fence();
        // a7 as a stage: P[j] = sum_t g[t] A[(j - t) mod n]; lane owns j = 8 l .. 8 l + 7 plus an 8-bin halo
        float a[16];
        {
            const float4 *f4 = reinterpret_cast<const float4 *>(fbuf);
            const int prev = 3 * ((l + 63) & 63);   // float4 index of bin 8 l - 8 (mod 512) under dp_fidx
            const float4 h0 = f4[prev], h1 = f4[prev + 1], c0 = f4[3 * l], c1 = f4[3 * l + 1];
            a[0] = h0.x; a[1] = h0.y; a[2] = h0.z; a[3] = h0.w;
            a[4] = h1.x; a[5] = h1.y; a[6] = h1.z; a[7] = h1.w;
            a[8] = c0.x; a[9] = c0.y; a[10] = c0.z; a[11] = c0.w;
            a[12] = c1.x; a[13] = c1.y; a[14] = c1.z; a[15] = c1.w;
        }
        if (do_dump && dump.abs2) {
#pragma unroll
            for (int u = 0; u < 8; u++) dump.abs2[(size_t)gate * DP_N + 8 * l + u] = a[8 + u];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            float p = 0.f;
#pragma unroll
            for (int t = 0; t < TAPS; t++) p = fmaf(taps.g[t], a[8 + u - t], p);
            if (do_dump && dump.pow) dump.pow[(size_t)gate * DP_N + 8 * l + u] = p;
        }
        if (do_dump && dump.rowsum && l == 0) dump.rowsum[gate] = S;
        wave_lds_fence();   // conv reads done before the buffer's next use
    }
    return S;
}

// a9: reflectivity (rpv2.cu:199-213): z = (gate*k_rr)^2 * k_cal * S_hh in double, rounded once
__device__ __forceinline__ void reflectivity_store(float *out2, int gate, float s_hh, float s_vv, float k_rr, float k_cal,
                                                   unsigned *frames = nullptr, int gates = 0, unsigned hdr = 0)
{
    const double rng = (double)gate * (double)k_rr;
    const float z = (float)(rng * rng * (double)k_cal * (double)s_hh);
    const float zdb = 10.f * log10f(z);
    const float zdr = 10.f * (log10f(s_hh) - log10f(s_vv));
    *reinterpret_cast<float2 *>(out2) = make_float2(zdb, zdr);
    if (frames) {   // wire-ready: the same bits, big-endian, planar (read_single.cc:510-520, rpv2.cu:631-661)
        frames[1 + gate] = __builtin_bswap32(__builtin_bit_cast(unsigned, zdb));
        frames[1 + gates + 1 + gate] = __builtin_bswap32(__builtin_bit_cast(unsigned, zdr));
        if (gate == 0) {   // the header words of the two frames travel with gate 0
            frames[0] = hdr;
            frames[1 + gates] = hdr;
        }
    }
}

// gridDim.y = n_sectors: one sector per workgroup row.  The gated repeat of a batch (a launch that almost never has
// anything to do: wrp_engine.hip, submit_fused_piece) is queued with a few rows only, which walk the sectors.
template <bool DUMP, int TAPS>
__global__ __launch_bounds__(DP_WAVES * 64) void doppler_pass_512(
    const float2 *__restrict__ mid,  // [S][2][gates][512]
    float *__restrict__ out,         // [S][gates][2]
    const float2 *__restrict__ tw,   // [DP_TW_ELEMS] exp(+2 pi i k / 512), arranged (doppler_twiddle_index)
    int gates, int n_sectors, MaTaps taps, float k_rr, float k_cal, DumpPtrs dump, const unsigned *gate_word)
{
    __shared__ __attribute__((aligned(16))) float2 lds[DP_WAVES][DP_ELEMS];
    __shared__ __attribute__((aligned(16))) float2 s_tw[DP_TW_ELEMS];
    if (gate_closed(gate_word)) return;
    const int w = wave_id(), l = threadIdx.x & 63;
    const int gate = blockIdx.x * DP_WAVES + w;
    bool tables = false;
#pragma unroll 1
    for (int sec = blockIdx.y; sec < n_sectors; sec += gridDim.y) {
        cf x[2][8];                      // both rows in flight before any arithmetic
#pragma unroll
        for (int ch = 0; ch < 2; ch++)
            doppler_load_row<AUX_NT>(mid + (((size_t)sec * 2 + ch) * gates + gate) * DP_N, l, x[ch]);
        if (!tables) {                   // (workgroup-uniform)
            doppler_twiddles_to_lds(s_tw, tw, threadIdx.x, DP_WAVES * 64);
            __syncthreads();
            tables = true;
        }
        float S[2];
#pragma unroll
        for (int ch = 0; ch < 2; ch++)
            S[ch] = doppler_row<DUMP, TAPS>(x[ch], lds[w], s_tw, taps, l, gate, DUMP && dump.channel == ch && sec == 0, dump);
        if (l == 0) {
            unsigned hdr;
            unsigned *frames = sector_frames(dump, sec, gates, hdr);
            reflectivity_store(&out[((size_t)sec * gates + gate) * 2], gate, S[0], S[1], k_rr, k_cal, frames, gates, hdr);
        }
    }
}

// =============================================================================================
// wire decode: the sector as it arrives -- 12 bytes per sample, hhI hhQ vvI vvQ
// vhI vhQ as big-endian int16 (sector.cpp:52-62) -- to the planar fp32 block [C][m][n] that
// read_matrix builds on the CPU (rpv2.cu:369-383).  One sample per thread: one 12-byte load,
// `channels` coalesced 8-byte stores.  Integer -> float is exact, so this is bit-identical to
// Sector::fromByteArray + the scatter loop.
// WB = 8: the same sample without its VH pair (hhI hhQ vvI vvQ), which the feeder drops in the copy it makes anyway
// (WRP_FLAG_WIRE_8, include/wrp.h): no output reads VH (rpv2.cu:199-213); a third plane of the block is left as it is.
// =============================================================================================
template <int WB>
__global__ __launch_bounds__(256) void decode_wire(const unsigned *__restrict__ raw,   // [n_sectors][count][WB / 4] dwords
                                                    float2 *__restrict__ iq,            // [n_sectors][channels][count]
                                                    int count, int channels, int n_sectors, const unsigned *gate)
{
    constexpr int D = WB / 4;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= count || gate_closed(gate)) return;
    for (int sec = blockIdx.y; sec < n_sectors; sec += gridDim.y) {     // gridDim.y = n_sectors, or a few rows (gated repeat)
        const unsigned *r = raw + (size_t)sec * count * D;
        float2 *q = iq + (size_t)sec * channels * count;
        unsigned w[D];
#pragma unroll
        for (int c = 0; c < D; c++) w[c] = __builtin_bswap32(r[D * t + c]);
#pragma unroll
        for (int c = 0; c < D; c++)
            if (c < channels)
                q[(size_t)c * count + t] = make_float2((float)(short)(w[c] >> 16), (float)(short)(w[c] & 0xffffu));
    }
}

} // namespace wrp
