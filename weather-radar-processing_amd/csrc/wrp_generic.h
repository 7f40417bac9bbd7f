// wrp_generic.h -- shape-generic (any power-of-two m <= 2048, n <= 1024) versions of the two
// passes.  They exist so that every sector shape of BASELINE.json's configs (config 5 is
// m = 2048, n = 128) and small test shapes run ON THE GPU through the same C ABI; they are plain
// radix-2 LDS FFTs and are not tuned.  The m = 1024 / n = 512 shape never takes this path.
// Stage semantics, dumps and the reflectivity formula are identical to wrp_kernels.h.
#pragma once
#include <hip/hip_runtime.h>

#include "wrp_kernels.h"

namespace wrp {

constexpr int GEN_TC = 8;          // columns per range workgroup
constexpr int GEN_THREADS = 256;

__device__ __forceinline__ int bitrev(int x, int bits) { return (int)(__brev((unsigned)x) >> (32 - bits)); }

// a2 + a3 for one (sector, channel, 8-column tile): in-place radix-2 DIF over m rows in LDS.
// After the last stage position p holds gate bitrev(p); gates < m/2 are stored.
template <bool DUMP>
__global__ __launch_bounds__(GEN_THREADS) void generic_range_pass(
    const float2 *__restrict__ iq, float2 *__restrict__ mid, RangeConsts rc, int m, int log2m, int n, int channels,
    DumpPtrs dump)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
    float2 *buf = reinterpret_cast<float2 *>(gsm);          // [m][GEN_TC]
    const int tiles = n / GEN_TC;
    int b = blockIdx.x;
    const int tile = b % tiles; b /= tiles;
    const int ch = b % 2;       b /= 2;
    const int sec = b;
    const float2 *src = iq + ((size_t)sec * channels + ch) * (size_t)m * n;
    float2 *dst = mid + ((size_t)sec * 2 + ch) * (size_t)(m / 2) * n;
    const bool do_dump = DUMP && dump.channel == ch && sec == 0;
    const int c = threadIdx.x % GEN_TC, col = tile * GEN_TC + c;
    const float wdc = rc.wd[col];
    for (int i = threadIdx.x / GEN_TC; i < m; i += GEN_THREADS / GEN_TC) {
        const float2 x = src[(size_t)i * n + col];
        const float w = rc.wr_c[i] * wdc;
        const float2 y = make_float2(x.x * w, x.y * w);
        if (do_dump && dump.hamm) dump.hamm[(size_t)i * n + col] = y;
        buf[i * GEN_TC + c] = y;
    }
    __syncthreads();
    for (int s = m / 2, st = 1; s >= 1; s >>= 1, st <<= 1) {     // twiddle W_m^{(b % s) * st}
        for (int bf = threadIdx.x / GEN_TC; bf < m / 2; bf += GEN_THREADS / GEN_TC) {
            const int lo = bf % s, i = (bf / s) * 2 * s + lo, j = i + s;
            const cf a = buf[i * GEN_TC + c], d = buf[j * GEN_TC + c];
            buf[i * GEN_TC + c] = cadd(a, d);
            buf[j * GEN_TC + c] = cmul(csub(a, d), rc.tw[(lo * st) & (m - 1)]);
        }
        __syncthreads();
    }
    for (int p = threadIdx.x / GEN_TC; p < m; p += GEN_THREADS / GEN_TC) {
        const int k = bitrev(p, log2m);
        const cf v = buf[p * GEN_TC + c];
        if (k < m / 2) dst[(size_t)k * n + col] = v;
        if (do_dump && dump.fft1) dump.fft1[(size_t)k * n + col] = v;
    }
}

// a4 .. a9 for one gate, one wave, both channels; radix-2 DIF with the +i exponent.
template <bool DUMP, int TAPS>
__global__ __launch_bounds__(64) void generic_doppler_pass(
    const float2 *__restrict__ mid, float *__restrict__ out, const float2 *__restrict__ tw, int gates, int n, int log2n,
    MaTaps taps, float k_rr, float k_cal, DumpPtrs dump)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
    float2 *buf = reinterpret_cast<float2 *>(gsm);            // [n]
    float *abuf = reinterpret_cast<float *>(gsm + (size_t)n * 8);   // [n]
    const int l = threadIdx.x, gate = blockIdx.x, sec = blockIdx.y;
    float S[2];
    for (int ch = 0; ch < 2; ch++) {
        const bool do_dump = DUMP && dump.channel == ch && sec == 0;
        const float2 *row = mid + (((size_t)sec * 2 + ch) * gates + gate) * (size_t)n;
        float sr = 0.f, si = 0.f;
        for (int j = l; j < n; j += 64) { const float2 x = row[j]; buf[j] = x; sr += x.x; si += x.y; }
        sr = wave_sum(sr) / (float)n;
        si = wave_sum(si) / (float)n;
        wave_lds_fence();
        for (int j = l; j < n; j += 64) { buf[j].x -= sr; buf[j].y -= si; }
        wave_lds_fence();
        for (int s = n / 2, st = 1; s >= 1; s >>= 1, st <<= 1) {
            for (int bf = l; bf < n / 2; bf += 64) {
                const int lo = bf % s, i = (bf / s) * 2 * s + lo, j = i + s;
                const cf a = buf[i], d = buf[j];
                buf[i] = cadd(a, d);
                buf[j] = cmul(csub(a, d), tw[(lo * st) & (n - 1)]);     // tw = exp(+2 pi i k / n)
            }
            wave_lds_fence();
        }
        // position bitrev(k) holds Z[k]; post-shift bin j is Z[(j + n/2) mod n]
        float part = 0.f;
        for (int j = l; j < n; j += 64) {
            const int k = (j + n / 2) & (n - 1);
            cf z = buf[bitrev(k, log2n)];
            if (do_dump && dump.noshift) dump.noshift[(size_t)gate * n + k] = make_float2(z.x, -z.y);
            if (j >= n - 2) z = make_float2(0.f, 0.f);
            if (do_dump && dump.fft2) dump.fft2[(size_t)gate * n + j] = z;
            const float a = fmaf(z.y, z.y, z.x * z.x);
            part += a;
            if (DUMP) abuf[j] = a;
            if (do_dump && dump.abs2) dump.abs2[(size_t)gate * n + j] = a;
        }
        wave_lds_fence();
        if (DUMP && do_dump && dump.pow)    // a7 as a stage only: its row sum is (sum of the taps) x (sum of |.|^2), see doppler_row
            for (int j = l; j < n; j += 64) {
                float p = 0.f;
#pragma unroll
                for (int t = 0; t < TAPS; t++) p = fmaf(taps.g[t], abuf[(j - t) & (n - 1)], p);
                dump.pow[(size_t)gate * n + j] = p;
            }
        S[ch] = wave_sum(part) * taps.sum;
        if (do_dump && dump.rowsum && l == 0) dump.rowsum[gate] = S[ch];
        wave_lds_fence();
    }
    if (l == 0) {
        unsigned hdr;
        unsigned *frames = sector_frames(dump, sec, gates, hdr);
        reflectivity_store(&out[((size_t)sec * gates + gate) * 2], gate, S[0], S[1], k_rr, k_cal, frames, gates, hdr);
    }
}

} // namespace wrp
