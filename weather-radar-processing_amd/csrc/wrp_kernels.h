// wrp_kernels.h -- the two fused HIP kernels of the per-sector chain (gfx950).
//
//   range_pass   : Hamming window (a2) + range FFT along i (a3), one workgroup per
//                  (sector, channel, 16-column tile); writes rows k < m/2 only.
//   doppler_pass : mean removal (a4), Doppler FFT + conj + shift + clip (a5), |.|^2 (a6),
//                  7-tap causal circular MA (a7), row sum (a8), Zdb/Zdr (a9); one wave
//                  per range gate, both polarisations in the same wave.
//
// Reference semantics: read.cc:133-345 / rpv2.cu:86-213,409-570 (see DESIGN.md §2 for the
// per-stage mapping).  No rocFFT/hipFFT: the FFTs are LDS-resident mixed-radix passes
// (16x8x8 for m = 1024, 8x8x8 for n = 512) built from fft_radix.h.
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
};

struct MaTaps { float g[9]; };

// ---------------------------------------------------------------------------------------------
// range pass, m = 1024 = 16 x 8 x 8; TCOLS = 16 columns per 512-thread workgroup (1 per CU) or
// 8 columns per 256-thread workgroup (2 per CU).
//
// In-place decimation-in-frequency over positions p of a column:
//   stage 1 (registers, straight from HBM): lane owns rows p0 + 64 r, r < 16   -> radix 16,
//           twiddle W_1024^{p0 k1}, result to LDS position k1*64 + p0
//   stage 2 (LDS): rows k1*64 + p1 + 8 r, r < 8  -> radix 8, twiddle W_64^{p1 k2}, in place
//   stage 3 (LDS): rows k1*64 + k2*8 + r, r < 8  -> radix 8; output row k = k1 + 16 k2 + 128 k3
// LDS image: [position][TCOLS columns] complex, plus one position of padding after
// every 8 positions so that stage 3's ds_read_b128 (lanes = column pairs x 8 k2) is
// bank-conflict free; stages 1 and 2 touch whole contiguous 1 KiB rows per wave-instruction.
// ---------------------------------------------------------------------------------------------
constexpr int RP_M = 1024;

template <int TCOLS>
struct RangeTile {
    static constexpr int CP = TCOLS / 2;              // column pairs (one float4) per row segment
    static constexpr int ROWS_PER_WAVE = 64 / CP;     // row segments one wave-instruction covers
    static constexpr int WAVES = 64 / ROWS_PER_WAVE;  // waves so that the block covers 64 rows
    static constexpr int THREADS = 64 * WAVES;        // 512 (16 columns) or 256 (8 columns)
    static constexpr int ROW_BYTES = TCOLS * 8;       // bytes of one position in LDS
    static constexpr int BLK_BYTES = 9 * ROW_BYTES;   // 8 positions + one position of padding
    static constexpr int LDS_BYTES = (RP_M / 8) * BLK_BYTES;   // 147456 / 73728
    static __device__ __forceinline__ int addr(int pos, int colpair)   // byte address of a float4
    {
        return (pos >> 3) * BLK_BYTES + (pos & 7) * ROW_BYTES + colpair * 16;
    }
};

template <int TCOLS, bool DUMP>
__global__ __launch_bounds__(RangeTile<TCOLS>::THREADS) void range_pass_1024(
    const float2 *__restrict__ iq,   // [S][C][1024][n]
    float2 *__restrict__ mid,        // [S][2][512][n]
    const float *__restrict__ wr_c,  // [1024]  range window * c
    const float *__restrict__ wd,    // [n]     Doppler window
    const float2 *__restrict__ tw,   // [1024]  exp(-2 pi i k / 1024)
    int n, int channels, DumpPtrs dump)
{
    typedef RangeTile<TCOLS> T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
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

    // ---- stage 1 ------------------------------------------------------------------------
    {
        const int w = tid >> 6, l = tid & 63;
        const int rowin = l / T::CP, cp = l % T::CP;
        const int p0 = w * T::ROWS_PER_WAVE + rowin;
        const int col0 = tile * TCOLS + cp * 2;
        float4 v[16];
#pragma unroll
        for (int r = 0; r < 16; r++)
            v[r] = *reinterpret_cast<const float4 *>(&src[(size_t)(p0 + 64 * r) * n + col0]);
        const float2 wdv = *reinterpret_cast<const float2 *>(&wd[col0]);
        cf a[16], c[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float wrow = wr_c[p0 + 64 * r];
            const float w0 = wrow * wdv.x, w1 = wrow * wdv.y;
            a[r] = make_float2(v[r].x * w0, v[r].y * w0);
            c[r] = make_float2(v[r].z * w1, v[r].w * w1);
        }
        if (do_dump && dump.hamm) {
#pragma unroll
            for (int r = 0; r < 16; r++)
                *reinterpret_cast<float4 *>(&dump.hamm[(size_t)(p0 + 64 * r) * n + col0]) =
                    make_float4(a[r].x, a[r].y, c[r].x, c[r].y);
        }
        fft16<-1>(a);
        fft16<-1>(c);
        *reinterpret_cast<float4 *>(smem + T::addr(p0, cp)) = make_float4(a[0].x, a[0].y, c[0].x, c[0].y);
#pragma unroll
        for (int k1 = 1; k1 < 16; k1++) {
            const cf t = tw[(p0 * k1) & (RP_M - 1)];
            const cf x = cmul(a[k1], t), y = cmul(c[k1], t);
            *reinterpret_cast<float4 *>(smem + T::addr(k1 * 64 + p0, cp)) = make_float4(x.x, x.y, y.x, y.y);
        }
    }
    __syncthreads();

    const int cp = tid % T::CP, q = (tid / T::CP) & 7, kb = tid / (T::CP * 8);
    // ---- stage 2 ------------------------------------------------------------------------
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const int k1 = kb + 8 * it, p1 = q;
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
            const cf t = tw[(16 * p1 * k2) & (RP_M - 1)];
            const cf x = cmul(a[k2], t), y = cmul(c[k2], t);
            *reinterpret_cast<float4 *>(smem + T::addr(k1 * 64 + p1 + 8 * k2, cp)) = make_float4(x.x, x.y, y.x, y.y);
        }
    }
    __syncthreads();

    // ---- stage 3 + store ----------------------------------------------------------------
    const int col0 = tile * TCOLS + cp * 2;
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const int k1 = kb + 8 * it, k2 = q;
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
        for (int k3 = 0; k3 < 4; k3++)   // rows k < m/2 only: the chain never reads the rest (rpv2.cu:502)
            *reinterpret_cast<float4 *>(&dst[(size_t)(k0 + 128 * k3) * n + col0]) =
                make_float4(a[k3].x, a[k3].y, c[k3].x, c[k3].y);
        if (do_dump && dump.fft1) {
#pragma unroll
            for (int k3 = 0; k3 < 8; k3++)
                *reinterpret_cast<float4 *>(&dump.fft1[(size_t)(k0 + 128 * k3) * n + col0]) =
                    make_float4(a[k3].x, a[k3].y, c[k3].x, c[k3].y);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// doppler pass, n = 512 = 8 x 8 x 8, one wave per range gate (both channels), 4 waves per block.
//
// Wave-private LDS, no workgroup barrier anywhere.  Position p of the in-place DIF lives at
// element p + (p >> 3) (one 8-byte pad per 8 elements): stage 1 writes, stage 2 and stage 3
// ds_read_b64 are then conflict free per 32-lane group.
// ---------------------------------------------------------------------------------------------
constexpr int DP_N = 512;
constexpr int DP_WAVES = 4;
constexpr int DP_ELEMS = DP_N + DP_N / 8;   // padded complex elements per wave buffer (576)

__device__ __forceinline__ int dp_idx(int pos) { return pos + (pos >> 3); }

__device__ __forceinline__ void wave_lds_fence()
{
    // LDS operations of one wave execute in issue order; this only stops the compiler from
    // moving accesses across the point where lanes exchange data.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <bool DUMP, int TAPS>
__global__ __launch_bounds__(DP_WAVES * 64) void doppler_pass_512(
    const float2 *__restrict__ mid,  // [S][2][gates][512]
    float *__restrict__ out,         // [S][gates][2]
    const float2 *__restrict__ tw,   // [512] exp(+2 pi i k / 512)
    int gates, MaTaps taps, float k_rr, float k_cal, DumpPtrs dump)
{
    __shared__ __attribute__((aligned(16))) float2 lds[DP_WAVES][DP_ELEMS];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int gate = blockIdx.x * DP_WAVES + w;
    const int sec = blockIdx.y;
    float2 *buf = lds[w];
    float *fbuf = reinterpret_cast<float *>(buf);

    // both rows in flight before any arithmetic
    cf x[2][8];
#pragma unroll
    for (int ch = 0; ch < 2; ch++) {
        const float2 *row = mid + (((size_t)sec * 2 + ch) * gates + gate) * DP_N;
#pragma unroll
        for (int r = 0; r < 8; r++) x[ch][r] = row[l + 64 * r];
    }
    // per-lane twiddles, shared by both channels
    cf t1[8], t2[8];
#pragma unroll
    for (int k = 1; k < 8; k++) {
        t1[k] = tw[(l * k) & (DP_N - 1)];            // W_512^{l k}
        t2[k] = tw[(8 * (l & 7) * k) & (DP_N - 1)];  // W_64^{p1 k}
    }

    float S[2];
#pragma unroll
    for (int ch = 0; ch < 2; ch++) {
        cf(&v)[8] = x[ch];
        const bool do_dump = DUMP && dump.channel == ch && sec == 0;
        // a4: mean over the row, subtract (rpv2.cu:434-439)
        float sr = 0.f, si = 0.f;
#pragma unroll
        for (int r = 0; r < 8; r++) { sr += v[r].x; si += v[r].y; }
        sr = wave_sum(sr) * (1.0f / DP_N);
        si = wave_sum(si) * (1.0f / DP_N);
#pragma unroll
        for (int r = 0; r < 8; r++) { v[r].x -= sr; v[r].y -= si; }

        // a5: Z[k] = sum_j (x_j - mu) exp(+2 pi i j k / n)   (= conj . FFT . conj)
        // stage 1: lane l owns j = l + 64 r
        fft8<+1>(v);
        buf[dp_idx(l)] = v[0];
#pragma unroll
        for (int k1 = 1; k1 < 8; k1++) buf[dp_idx(k1 * 64 + l)] = cmul(v[k1], t1[k1]);
        wave_lds_fence();
        // stage 2: lane = p1 + 8 k1 owns positions k1*64 + p1 + 8 r
        {
            const int p1 = l & 7, k1 = l >> 3;
#pragma unroll
            for (int r = 0; r < 8; r++) v[r] = buf[dp_idx(k1 * 64 + p1 + 8 * r)];
            fft8<+1>(v);
            buf[dp_idx(k1 * 64 + p1)] = v[0];
#pragma unroll
            for (int k2 = 1; k2 < 8; k2++) buf[dp_idx(k1 * 64 + p1 + 8 * k2)] = cmul(v[k2], t2[k2]);
        }
        wave_lds_fence();
        // stage 3: lane = k2 + 8 k1 owns positions k1*64 + k2*8 + r; output k = k1 + 8 k2 + 64 k3
        const int k2 = l & 7, k1 = l >> 3;
        const int klo = k1 + 8 * k2;
#pragma unroll
        for (int r = 0; r < 8; r++) v[r] = buf[dp_idx(k1 * 64 + k2 * 8 + r)];
        fft8<+1>(v);
        wave_lds_fence();   // everyone has read before the buffer is reused for |.|^2

        if (do_dump && dump.noshift) {   // reference dumps the FFT output before the final conj
#pragma unroll
            for (int k3 = 0; k3 < 8; k3++)
                dump.noshift[(size_t)gate * DP_N + klo + 64 * k3] = make_float2(v[k3].x, -v[k3].y);
        }
        // shift (swap halves: j = k + n/2 mod n), clip post-shift bins n-1, n-2, |.|^2
#pragma unroll
        for (int k3 = 0; k3 < 8; k3++) {
            const int j = ((k3 + 4) & 7) * 64 + klo;
            cf z = v[k3];
            if (j >= DP_N - 2) z = make_float2(0.f, 0.f);
            if (do_dump && dump.fft2) dump.fft2[(size_t)gate * DP_N + j] = z;
            fbuf[j] = z.x * z.x + z.y * z.y;
        }
        wave_lds_fence();
        // a7: P[j] = sum_t g[t] A[(j - t) mod n]; lane owns j = 8 l .. 8 l + 7
        float a[16];
        {
            const float4 *f4 = reinterpret_cast<const float4 *>(fbuf);
            const int base = (2 * l + 126) & 127;   // float4 index of element 8 l - 8 (mod 512)
            const float4 h0 = f4[base], h1 = f4[(base + 1) & 127], c0 = f4[2 * l], c1 = f4[2 * l + 1];
            a[0] = h0.x; a[1] = h0.y; a[2] = h0.z; a[3] = h0.w;
            a[4] = h1.x; a[5] = h1.y; a[6] = h1.z; a[7] = h1.w;
            a[8] = c0.x; a[9] = c0.y; a[10] = c0.z; a[11] = c0.w;
            a[12] = c1.x; a[13] = c1.y; a[14] = c1.z; a[15] = c1.w;
        }
        if (do_dump && dump.abs2) {
#pragma unroll
            for (int u = 0; u < 8; u++) dump.abs2[(size_t)gate * DP_N + 8 * l + u] = a[8 + u];
        }
        float part = 0.f;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            float p = 0.f;
#pragma unroll
            for (int t = 0; t < TAPS; t++) p = fmaf(taps.g[t], a[8 + u - t], p);
            if (do_dump && dump.pow) dump.pow[(size_t)gate * DP_N + 8 * l + u] = p;
            part += p;
        }
        // a8: row sum
        S[ch] = wave_sum(part);
        if (do_dump && dump.rowsum && l == 0) dump.rowsum[gate] = S[ch];
        wave_lds_fence();   // conv reads done before the next channel's stage 1 writes
    }
    // a9: reflectivity (rpv2.cu:199-213): z = (gate*k_rr)^2 * k_cal * S_hh in double, rounded once
    if (l == 0) {
        const double rng = (double)gate * (double)k_rr;
        const float z = (float)(rng * rng * (double)k_cal * (double)S[0]);
        const float zdb = 10.f * log10f(z);
        const float zdr = 10.f * (log10f(S[0]) - log10f(S[1]));
        *reinterpret_cast<float2 *>(&out[((size_t)sec * gates + gate) * 2]) = make_float2(zdb, zdr);
    }
}

} // namespace wrp
