// tilebench.hip -- the range tile (16 columns x 1024 rows -> gates < 512: a2 + a3 of one tile task of the fused launch) as a
// stand-alone kernel on the whole chip, in the forms VERDICT r03 (next 1a) asks to be MEASURED instead of counted:
//
//   P    the present tile of fused_chain_1024x512: 1024 = 16 x 8 x 8, 512 threads, two k1-groups through a 72 KiB image, four
//        barriers, 128 VGPRs (the device functions of csrc/wrp_fused.h themselves, with the launch's request schedule);
//        one or two such workgroups per CU
//   R32  1024 = 32 x 32: ONE LDS exchange (132 KiB image), second stage pruned to its lower sixteen outputs, two barriers;
//        256 VGPRs = two waves per SIMD: 512 threads (one butterfly per lane and stage) or 256 threads (two), ONE workgroup
//        per CU (the image leaves no room for a second)
//   (R32 at four waves per SIMD does not exist: a 128-register wave cannot hold a radix-32 butterfly (64 + ~30 registers)
//    beside the next tile's 64, and with the waiting outputs parked in LDS the image is 132 KiB + tables = 144 KiB per tile
//    workgroup, where the CU has 160 KiB for the tile workgroup AND the row workgroup's 36 KiB of wave buffers.)
//
// Every workgroup transforms `tiles` tiles back to back: input from HBM (distinct tiles, non-temporal, the next tile
// requested while the current one is transformed) or dropped (zero-record descriptor), output stored into an L2-resident
// slot as in the launch.  Reported: us per tile per workgroup, tiles per us per CU, and the sector-equivalent (a sector is 64
// tiles).  --check compares both forms against a double-precision DFT on the host.
//
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -I weather-radar-processing_amd/csrc
//              -o build/tools/tilebench tools/tilebench.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "wrp_fused.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

namespace wrp {

// ------------------------------------------------------------------------------------------------------------------
// radix 32 in registers, natural order in, result X[k] at v[(k >> 2) + 8 (k & 3)] (r32_at): 32 = 4 x 8,
//   X[k' + 4 k''] = sum_a W8^{a k''} [ W32^{a k'} sum_b x[a + 8 b] W4^{b k'} ]
// ------------------------------------------------------------------------------------------------------------------
__device__ constexpr float C32[32] = {1.000000000f, 0.980785280f, 0.923879533f, 0.831469612f, 0.707106781f, 0.555570233f, 0.382683432f, 0.195090322f,
                                      0.000000000f, -0.195090322f, -0.382683432f, -0.555570233f, -0.707106781f, -0.831469612f, -0.923879533f, -0.980785280f,
                                      -1.000000000f, -0.980785280f, -0.923879533f, -0.831469612f, -0.707106781f, -0.555570233f, -0.382683432f, -0.195090322f,
                                      0.000000000f, 0.195090322f, 0.382683432f, 0.555570233f, 0.707106781f, 0.831469612f, 0.923879533f, 0.980785280f};
__device__ constexpr float S32[32] = {0.000000000f, 0.195090322f, 0.382683432f, 0.555570233f, 0.707106781f, 0.831469612f, 0.923879533f, 0.980785280f,
                                      1.000000000f, 0.980785280f, 0.923879533f, 0.831469612f, 0.707106781f, 0.555570233f, 0.382683432f, 0.195090322f,
                                      0.000000000f, -0.195090322f, -0.382683432f, -0.555570233f, -0.707106781f, -0.831469612f, -0.923879533f, -0.980785280f,
                                      -1.000000000f, -0.980785280f, -0.923879533f, -0.831469612f, -0.707106781f, -0.555570233f, -0.382683432f, -0.195090322f};
template <int SIGN, int E>
__device__ __forceinline__ cf mul_w32(cf a)   // a * exp(SIGN * 2 pi i E / 32)
{
    constexpr int e = E & 31;
    if (e == 0) return a;
    if (e == 8) return mul_si<SIGN>(a);
    if (e == 16) return make_float2(-a.x, -a.y);
    if (e == 24) return mul_si<-SIGN>(a);
    if (e == 4) return mul_w8_1<SIGN>(a);
    if (e == 12) return mul_w8_3<SIGN>(a);
    return cmul(a, make_float2(C32[e], (float)SIGN * S32[e]));
}
template <int SIGN, int KP>
__device__ __forceinline__ void r32_twiddle_row(cf (&v)[32])
{
    v[1 + 8 * KP] = mul_w32<SIGN, 1 * KP>(v[1 + 8 * KP]);
    v[2 + 8 * KP] = mul_w32<SIGN, 2 * KP>(v[2 + 8 * KP]);
    v[3 + 8 * KP] = mul_w32<SIGN, 3 * KP>(v[3 + 8 * KP]);
    v[4 + 8 * KP] = mul_w32<SIGN, 4 * KP>(v[4 + 8 * KP]);
    v[5 + 8 * KP] = mul_w32<SIGN, 5 * KP>(v[5 + 8 * KP]);
    v[6 + 8 * KP] = mul_w32<SIGN, 6 * KP>(v[6 + 8 * KP]);
    v[7 + 8 * KP] = mul_w32<SIGN, 7 * KP>(v[7 + 8 * KP]);
}
template <int SIGN>
__device__ __forceinline__ void fft32_tail(cf (&v)[32])
{
    r32_twiddle_row<SIGN, 1>(v);
    r32_twiddle_row<SIGN, 2>(v);
    r32_twiddle_row<SIGN, 3>(v);
#pragma unroll
    for (int kp = 0; kp < 4; kp++) fft8<SIGN>(reinterpret_cast<cf(&)[8]>(v[8 * kp]));
}
template <int SIGN>
__device__ __forceinline__ void fft32(cf (&v)[32])
{
#pragma unroll
    for (int a = 0; a < 8; a++) fft4<SIGN>(v[a], v[a + 8], v[a + 16], v[a + 24]);
    fft32_tail<SIGN>(v);
}
// fft32 of (w[r] s) v[r]: the window rides on the first level (fft4_scaled of fft_radix.h)
template <int SIGN>
__device__ __forceinline__ void fft32_scaled(cf (&v)[32], const float (&w)[32], float s)
{
#pragma unroll
    for (int a = 0; a < 8; a++) fft4_scaled<SIGN>(v[a], v[a + 8], v[a + 16], v[a + 24], w[a] * s, w[a + 8] * s, w[a + 16] * s, w[a + 24] * s);
    fft32_tail<SIGN>(v);
}
__device__ __forceinline__ constexpr int r32_at(int k) { return (k >> 2) + 8 * (k & 3); }

// LDS of the 32 x 32 tile: [32 k1][32 p0][16 columns] complex, 128 bytes of padding per k1 (a ds_read_b64 of stage 2 covers
// two k1 x 16 columns per half wave: the padding puts them on disjoint halves of the 64 banks), window, twiddles [p0][k1]
struct TileR32 {
    static constexpr int ROW = 128, K1S = 32 * ROW + 128, IMG = 32 * K1S;          // 135168
    static constexpr int OFF_WR = IMG, OFF_TW = OFF_WR + RP_M * 4, LDS_BYTES = OFF_TW + RP_M * 8;   // 147456
};

template <int THREADS, bool LOAD>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(1, 2))) void tile_r32(
    const float2 *__restrict__ iq, float2 *slots, RangeConsts rc, int tiles, int n_sc, float2 *check, int same)
{
    typedef TileR32 T;
    constexpr int ITEMS = 512 / THREADS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    // as in the launch: the workgroups of one XCD (blockIdx & 7 under round-robin placement) read ONE matrix per step, its 32
    // column tiles spread over them and rotated from step to step; SAME = 1: every step the same matrix (cache-served input)
    auto tile_src = [&](int t) { return iq + (size_t)(((blockIdx.x & 7) + 8 * (same ? 0 : t)) % n_sc) * RP_M * DP_N; };
    auto tile_col = [&](int t) { return (((blockIdx.x >> 3) + t) & 31) * 16; };
    cf v[ITEMS][32];
    float wdc[ITEMS];
    auto load = [&](int t, bool valid) {
        const rsrc_t rs = make_rsrc(tile_src(t), valid ? (unsigned)RP_M * DP_N * 8u : 0u);
        const rsrc_t rw = make_rsrc(rc.wd, (unsigned)DP_N * 4u);
#pragma unroll
        for (int it = 0; it < ITEMS; it++) {
            int lt = tid;
            asm volatile("" : "+v"(lt));
            const int c = lt & 15, p0 = (lt >> 4) + (THREADS / 16) * it;
            const int voff = (p0 * DP_N + tile_col(t) + c) * 8;
#pragma unroll
            for (int r = 0; r < 32; r++) v[it][r] = buf_load_f2<AUX_NT>(rs, voff, 32 * r * DP_N * 8);
            wdc[it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, (tile_col(t) + c) * 4, 0, 0));
        }
    };
    load(0, LOAD && tiles > 0);
    for (int e = tid; e < RP_M; e += THREADS) {
        const int p0 = e >> 5, k1 = e & 31;
        *reinterpret_cast<float2 *>(smem + T::OFF_TW + e * 8) = rc.tw[(p0 * k1) & (RP_M - 1)];
        reinterpret_cast<float *>(smem + T::OFF_WR)[e] = rc.wr_c[e];
    }
    __syncthreads();
    float2 *mid = slots + (size_t)(blockIdx.x & 7) * FUSED_TEAM_ELEMS;
    const int store_col = ((blockIdx.x >> 3) & 31) * 16;
#pragma unroll 1
    for (int t = 0; t < tiles; t++) {
        // stage 1: window, radix 32 over rows p0 + 32 r, twiddle W_1024^{p0 k1} -> image [k1][p0][c]
#pragma unroll
        for (int it = 0; it < ITEMS; it++) {
            int lt = tid;
            asm volatile("" : "+v"(lt));
            const int c = lt & 15, p0 = (lt >> 4) + (THREADS / 16) * it;
            const float *s_wr = reinterpret_cast<const float *>(smem + T::OFF_WR) + p0;
            float wr[32];
#pragma unroll
            for (int r = 0; r < 32; r++) wr[r] = s_wr[32 * r];
            fft32_scaled<-1>(v[it], wr, wdc[it]);
            unsigned char *dst = smem + p0 * T::ROW + c * 8;
            const unsigned char *tw = smem + T::OFF_TW + p0 * 32 * 8;
            *reinterpret_cast<float2 *>(dst) = v[it][r32_at(0)];
#pragma unroll
            for (int k1 = 1; k1 < 32; k1++)
                *reinterpret_cast<float2 *>(dst + k1 * T::K1S) = cmul(v[it][r32_at(k1)], *reinterpret_cast<const float2 *>(tw + k1 * 8));
        }
        __syncthreads();
        load(t + 1 < tiles ? t + 1 : 0, LOAD && t + 1 < tiles);   // v is free: the next tile flies during stage 2
        // stage 2: radix 32 over p0 for (k1, c); gates k1 + 32 k2, k2 < 16
#pragma unroll
        for (int it = 0; it < ITEMS; it++) {
            int lt = tid;
            asm volatile("" : "+v"(lt));
            const int c = lt & 15, k1 = (lt >> 4) + (THREADS / 16) * it;
            const unsigned char *src = smem + k1 * T::K1S + c * 8;
            cf a[32];
#pragma unroll
            for (int p = 0; p < 32; p++) a[p] = *reinterpret_cast<const float2 *>(src + p * T::ROW);
            fft32<-1>(a);
            const rsrc_t rd = make_rsrc(mid, (unsigned)FUSED_SLOT_ROWS * DP_N * 8u);
            const int voff = ((k1 & 7) * DP_N + store_col + c) * 8 + (k1 >> 3) * 64 * DP_N * 8;   // some row < 256 per gate: 128-byte lines as in the launch
#pragma unroll
            for (int k2 = 0; k2 < 16; k2++) {
                v2f o;
                o.x = a[r32_at(k2)].x; o.y = a[r32_at(k2)].y;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, o), rd, voff + (k2 & 7) * 8 * DP_N * 8, 0, 0);
                if (check && blockIdx.x == 0 && t == 0) check[(k1 + 32 * k2) * 16 + c] = a[r32_at(k2)];
            }
        }
        __syncthreads();
    }
}

// the present tile: the tile member's loop of fused_chain_1024x512 without the team protocol (no looks, no flags)
template <bool LOAD, bool CHECK = false>
__global__ __launch_bounds__(FUSED_THREADS, 4) __attribute__((amdgpu_waves_per_eu(4, 4))) void tile_p(
    const float2 *__restrict__ iq, float2 *slots, RangeConsts rc, int tiles, int n_sc, float2 *check, int same)
{
    typedef FusedTile T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, w = wave_id(), l = tid & 63;
    auto tile_src = [&](int t) { return iq + (size_t)(((blockIdx.x & 7) + 8 * (same ? 0 : t)) % n_sc) * RP_M * DP_N; };
    auto tile_col = [&](int t) { return (((blockIdx.x >> 3) + t) & 31) * 16; };
    float4 v[16];
    float2 wdv;
    fused_tile_load<0>(tile_src(0), tile_col(0), rc.wd, v, wdv, LOAD && tiles > 0);
    fused_tile_load<1>(tile_src(0), tile_col(0), rc.wd, v, wdv, LOAD && tiles > 0);
    fused_tile_load<2>(tile_src(0), tile_col(0), rc.wd, v, wdv, LOAD && tiles > 0);
    fused_tile_load<3>(tile_src(0), tile_col(0), rc.wd, v, wdv, LOAD && tiles > 0);
    for (int e = tid; e < RP_M; e += FUSED_THREADS) {
        const int p0 = e >> 4, k1 = e & 15;
        *reinterpret_cast<float2 *>(smem + T::tw1_addr(p0, k1)) = rc.tw[(p0 * k1) & (RP_M - 1)];
        reinterpret_cast<float *>(smem + T::OFF_WR)[e] = rc.wr_c[e];
    }
    if (tid < 64) *reinterpret_cast<float2 *>(smem + T::tw2_addr(tid >> 3, tid & 7)) = rc.tw[(16 * (tid >> 3) * (tid & 7)) & (RP_M - 1)];
    __syncthreads();
    float2 *mid = slots + (size_t)(blockIdx.x & 7) * FUSED_TEAM_ELEMS;
    const int store_col = ((blockIdx.x >> 3) & 31) * 16;
    auto dump = [&](int t, int group, const cf (&o)[2][4]) {
        if (CHECK && check && blockIdx.x == 0 && t == 0) {
#pragma unroll
            for (int it = 0; it < 2; it++)
#pragma unroll
                for (int k3 = 0; k3 < 4; k3++) check[((w + 8 * group) + 16 * ((l >> 4) + 4 * it) + 128 * k3) * 16 + (l & 15)] = o[it][k3];
        }
    };
#pragma unroll 1
    for (int t = 0; t < tiles; t++) {
        cf ga[8], gc[8];
        FusedStage1Tables s1t;
        fused_stage1_tables(smem, s1t);
        fused_stage1<0>(smem, v, wdv, s1t, ga);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        fused_stage1<1>(smem, v, wdv, s1t, gc);
        __syncthreads();                    // A1
        const float2 *next = tile_src(t + 1 < tiles ? t + 1 : 0);
        cf o[2][4];
        const int voff_next = fused_tile_voff(tile_col(t + 1));
#define TB_L1(R) fused_tile_load1<R>(next, voff_next, rc.wd, v, wdv, LOAD && t + 1 < tiles)
        TB_L1(0); TB_L1(8);
        fused_stage2_item<0>(smem);
        TB_L1(4);
        fused_stage2_item<1>(smem);
        TB_L1(12);
        fused_stage3_item<0>(smem, o);
        TB_L1(1);
        fused_stage3_item<1>(smem, o);
        TB_L1(9);
        __syncthreads();                    // A2
        fused_store(mid, store_col, 0, o);
        dump(t, 0, o);
        __builtin_amdgcn_sched_barrier(0);
        TB_L1(5); TB_L1(13); TB_L1(2); TB_L1(10);
        fused_group1_to_lds(smem, ga, gc);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __syncthreads();                    // A3
        TB_L1(6);
        fused_stage2_item<0>(smem);
        TB_L1(14);
        fused_stage2_item<1>(smem);
        TB_L1(3);
        fused_stage3_item<0>(smem, o);
        TB_L1(11);
        fused_stage3_item<1>(smem, o);
        TB_L1(7); TB_L1(15);
#undef TB_L1
        __syncthreads();                    // A4
        fused_store(mid, store_col, 1, o);
        dump(t, 1, o);
    }
}

// the input stream of the P form by itself: the same requests (sixteen 1 KiB loads per wave and tile, non-temporal), nothing else
template <int DEPTH>   // tiles in flight per workgroup: 1 (one set of 64 registers, as the launch) or 2
__global__ __launch_bounds__(FUSED_THREADS, 4) void tile_stream(const float2 *__restrict__ iq, float *sink, RangeConsts rc, int tiles, int n_sc, int same)
{
    auto tile_src = [&](int t) { return iq + (size_t)(((blockIdx.x & 7) + 8 * (same ? 0 : t)) % n_sc) * RP_M * DP_N; };
    auto tile_col = [&](int t) { return (((blockIdx.x >> 3) + t) & 31) * 16; };
    float4 v[DEPTH][16];
    float2 wdv;
    float acc = 0.f;
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
        fused_tile_load<0>(tile_src(d), tile_col(d), rc.wd, v[d], wdv, d < tiles);
        fused_tile_load<1>(tile_src(d), tile_col(d), rc.wd, v[d], wdv, d < tiles);
        fused_tile_load<2>(tile_src(d), tile_col(d), rc.wd, v[d], wdv, d < tiles);
        fused_tile_load<3>(tile_src(d), tile_col(d), rc.wd, v[d], wdv, d < tiles);
    }
#pragma unroll 1
    for (int t = 0; t < tiles; t += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
#pragma unroll
            for (int r = 0; r < 16; r++) acc += v[d][r].x + v[d][r].w;
            const int nt = t + d + DEPTH;
            fused_tile_load<0>(tile_src(nt < tiles ? nt : 0), tile_col(nt), rc.wd, v[d], wdv, nt < tiles);
            fused_tile_load<1>(tile_src(nt < tiles ? nt : 0), tile_col(nt), rc.wd, v[d], wdv, nt < tiles);
            fused_tile_load<2>(tile_src(nt < tiles ? nt : 0), tile_col(nt), rc.wd, v[d], wdv, nt < tiles);
            fused_tile_load<3>(tile_src(nt < tiles ? nt : 0), tile_col(nt), rc.wd, v[d], wdv, nt < tiles);
        }
    }
    if (acc == 123.456f) sink[threadIdx.x] = acc + wdv.x;
}

} // namespace wrp

int main(int argc, char **argv)
{
    using namespace wrp;
    int tiles = 192;
    bool do_check = true;
    const char *only = nullptr;      // --only SUBSTRING: time the variants whose line contains it (PMC passes)
    for (int a = 1; a < argc; a++) {
        if (!strcmp(argv[a], "--only") && a + 1 < argc) only = argv[++a];
        if (!strcmp(argv[a], "--tiles") && a + 1 < argc) tiles = atoi(argv[++a]);
        if (!strcmp(argv[a], "--no-check")) do_check = false;
    }
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int n_sc = 128;                                   // 128 sector-channels of 4 MiB: 512 MiB of input >> the Infinity Cache
    const size_t sc_elems = (size_t)RP_M * DP_N;
    std::vector<float2> h_in(sc_elems);
    srand(5);
    for (auto &x : h_in) x = make_float2((float)(rand() % 32768 - 16384), (float)(rand() % 32768 - 16384));
    float2 *d_in, *d_slots, *d_check;
    CK(hipMalloc(&d_in, sizeof(float2) * sc_elems * n_sc));
    for (int s = 0; s < n_sc; s++) CK(hipMemcpy(d_in + s * sc_elems, h_in.data(), sizeof(float2) * sc_elems, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_slots, sizeof(float2) * FUSED_TEAM_ELEMS * 8));
    CK(hipMalloc(&d_check, sizeof(float2) * 512 * 16));
    // constants as wrp_engine.hip makes them
    std::vector<float> wr(RP_M), wd(DP_N);
    {
        double pr = 0, pd = 0;
        for (int i = 0; i < RP_M; i++) pr += std::pow(0.53836 - 0.46164 * std::cos(2 * M_PI * i / (RP_M - 1)), 2.0);
        for (int j = 0; j < DP_N; j++) pd += std::pow(0.53836 - 0.46164 * std::cos(2 * M_PI * j / (DP_N - 1)), 2.0);
        pr /= RP_M; pd /= DP_N;
        const double c = (-1 / (16383.5 * RP_M * DP_N * std::sqrt(50.0))) / std::sqrt(pr * pd);
        for (int i = 0; i < RP_M; i++) wr[i] = (float)((0.53836 - 0.46164 * std::cos(2 * M_PI * i / (RP_M - 1))) * c);
        for (int j = 0; j < DP_N; j++) wd[j] = (float)(0.53836 - 0.46164 * std::cos(2 * M_PI * j / (DP_N - 1)));
    }
    std::vector<float2> tw(RP_M);
    for (int k = 0; k < RP_M; k++) tw[k] = make_float2((float)std::cos(2 * M_PI * k / RP_M), (float)-std::sin(2 * M_PI * k / RP_M));
    float *d_wr, *d_wd;
    float2 *d_tw;
    CK(hipMalloc(&d_wr, 4 * RP_M)); CK(hipMalloc(&d_wd, 4 * DP_N)); CK(hipMalloc(&d_tw, 8 * RP_M));
    CK(hipMemcpy(d_wr, wr.data(), 4 * RP_M, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_wd, wd.data(), 4 * DP_N, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tw, tw.data(), 8 * RP_M, hipMemcpyHostToDevice));
    const RangeConsts rc{d_wr, d_wd, d_tw};
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&tile_p<true>), hipFuncAttributeMaxDynamicSharedMemorySize, FusedTile::LDS_BYTES));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&tile_p<false>), hipFuncAttributeMaxDynamicSharedMemorySize, FusedTile::LDS_BYTES));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&tile_p<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, FusedTile::LDS_BYTES));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&tile_r32<512, true>), hipFuncAttributeMaxDynamicSharedMemorySize, TileR32::LDS_BYTES));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&tile_r32<512, false>), hipFuncAttributeMaxDynamicSharedMemorySize, TileR32::LDS_BYTES));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&tile_r32<256, true>), hipFuncAttributeMaxDynamicSharedMemorySize, TileR32::LDS_BYTES));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&tile_r32<256, false>), hipFuncAttributeMaxDynamicSharedMemorySize, TileR32::LDS_BYTES));

    if (do_check) {   // tile 0 (columns 0..15 of sector-channel 0) against a double-precision DFT
        std::vector<double> ref_re(512 * 16), ref_im(512 * 16);
        for (int c = 0; c < 16; c++) {
            std::vector<double> xr(RP_M), xi(RP_M);
            for (int i = 0; i < RP_M; i++) {
                const double wgt = (double)wr[i] * (double)wd[c];
                xr[i] = h_in[(size_t)i * DP_N + c].x * wgt;
                xi[i] = h_in[(size_t)i * DP_N + c].y * wgt;
            }
            for (int k = 0; k < 512; k++) {
                double sr = 0, si = 0;
                for (int i = 0; i < RP_M; i++) {
                    const double ang = -2 * M_PI * ((i * k) & (RP_M - 1)) / RP_M, cr = std::cos(ang), ci = std::sin(ang);
                    sr += xr[i] * cr - xi[i] * ci;
                    si += xr[i] * ci + xi[i] * cr;
                }
                ref_re[k * 16 + c] = sr; ref_im[k * 16 + c] = si;
            }
        }
        std::vector<float2> got(512 * 16);
        auto compare = [&](const char *what) {
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(got.data(), d_check, sizeof(float2) * got.size(), hipMemcpyDeviceToHost));
            double num = 0, den = 0, worst = 0;
            for (size_t e = 0; e < got.size(); e++) {
                const double dr = got[e].x - ref_re[e], di = got[e].y - ref_im[e];
                num += dr * dr + di * di;
                den += ref_re[e] * ref_re[e] + ref_im[e] * ref_im[e];
            }
            for (int k = 0; k < 512; k++) {
                double rowmax = 0;
                for (int c = 0; c < 16; c++) rowmax = std::max(rowmax, std::hypot(ref_re[k * 16 + c], ref_im[k * 16 + c]));
                for (int c = 0; c < 16; c++) worst = std::max(worst, std::hypot(got[k * 16 + c].x - ref_re[k * 16 + c], got[k * 16 + c].y - ref_im[k * 16 + c]) / rowmax);
            }
            printf("check %-26s L2-rel %.3g, worst |err| / rowmax %.3g\n", what, std::sqrt(num / den), worst);
        };
        CK(hipMemset(d_check, 0, sizeof(float2) * 512 * 16));
        hipLaunchKernelGGL((tile_p<true, true>), dim3(8), dim3(512), FusedTile::LDS_BYTES, 0, d_in, d_slots, rc, 1, n_sc, d_check, 0);
        compare("P (16 x 8 x 8)");
        CK(hipMemset(d_check, 0, sizeof(float2) * 512 * 16));
        hipLaunchKernelGGL((tile_r32<512, true>), dim3(8), dim3(512), TileR32::LDS_BYTES, 0, d_in, d_slots, rc, 1, n_sc, d_check, 0);
        compare("R32 (32 x 32), 512 threads");
        CK(hipMemset(d_check, 0, sizeof(float2) * 512 * 16));
        hipLaunchKernelGGL((tile_r32<256, true>), dim3(8), dim3(256), TileR32::LDS_BYTES, 0, d_in, d_slots, rc, 1, n_sc, d_check, 0);
        compare("R32 (32 x 32), 256 threads");
    }

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time_it = [&](const char *what, auto launch, int wgs_per_cu) {
        if (only && !strstr(what, only)) return;
        float best = 1e30f;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipEventRecord(e0));
            launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        const double us_tile = best * 1e3 / tiles;             // per workgroup
        const double tiles_per_us_cu = wgs_per_cu / us_tile;
        printf("%-44s %7.3f us per tile and workgroup, %6.4f tiles/us per CU, sector-equivalent (64 tiles over %d CUs) %6.3f us\n", what, us_tile,
               tiles_per_us_cu, cus, 64.0 / (tiles_per_us_cu * cus));
    };
    for (int variant = 0; variant < 3; variant++) {
        const int load = variant < 2, same = variant == 1;
        const char *sfx = variant == 0 ? "input from HBM" : variant == 1 ? "input cache-served" : "input dropped";
        if (load) {
            char nm[128];
            snprintf(nm, sizeof nm, "stream only, 1 tile in flight per WG, 1 WG/CU, %s", sfx);
            time_it(nm, [&] { hipLaunchKernelGGL(tile_stream<1>, dim3(cus), dim3(512), 0, 0, d_in, (float *)d_check, rc, tiles, n_sc, same); }, 1);
            snprintf(nm, sizeof nm, "stream only, 1 tile in flight per WG, 2 WG/CU, %s", sfx);
            time_it(nm, [&] { hipLaunchKernelGGL(tile_stream<1>, dim3(2 * cus), dim3(512), 0, 0, d_in, (float *)d_check, rc, tiles, n_sc, same); }, 2);
            snprintf(nm, sizeof nm, "stream only, 2 tiles in flight per WG, 1 WG/CU, %s", sfx);
            time_it(nm, [&] { hipLaunchKernelGGL(tile_stream<2>, dim3(cus), dim3(512), 0, 0, d_in, (float *)d_check, rc, tiles, n_sc, same); }, 1);
        }
        char name[128];
        snprintf(name, sizeof name, "P, 1 workgroup per CU, %s", sfx);
        time_it(name, [&] { if (load) hipLaunchKernelGGL(tile_p<true>, dim3(cus), dim3(512), FusedTile::LDS_BYTES, 0, d_in, d_slots, rc, tiles, n_sc, (float2 *)nullptr, same);
                            else hipLaunchKernelGGL(tile_p<false>, dim3(cus), dim3(512), FusedTile::LDS_BYTES, 0, d_in, d_slots, rc, tiles, n_sc, (float2 *)nullptr, same); }, 1);
        snprintf(name, sizeof name, "P, 2 workgroups per CU, %s", sfx);
        time_it(name, [&] { if (load) hipLaunchKernelGGL(tile_p<true>, dim3(2 * cus), dim3(512), FusedTile::LDS_BYTES, 0, d_in, d_slots, rc, tiles, n_sc, (float2 *)nullptr, same);
                            else hipLaunchKernelGGL(tile_p<false>, dim3(2 * cus), dim3(512), FusedTile::LDS_BYTES, 0, d_in, d_slots, rc, tiles, n_sc, (float2 *)nullptr, same); }, 2);
        snprintf(name, sizeof name, "R32, 512 threads (2 waves/SIMD), %s", sfx);
        time_it(name, [&] { if (load) hipLaunchKernelGGL((tile_r32<512, true>), dim3(cus), dim3(512), TileR32::LDS_BYTES, 0, d_in, d_slots, rc, tiles, n_sc, (float2 *)nullptr, same);
                            else hipLaunchKernelGGL((tile_r32<512, false>), dim3(cus), dim3(512), TileR32::LDS_BYTES, 0, d_in, d_slots, rc, tiles, n_sc, (float2 *)nullptr, same); }, 1);
        snprintf(name, sizeof name, "R32, 256 threads (1 wave/SIMD), %s", sfx);
        time_it(name, [&] { if (load) hipLaunchKernelGGL((tile_r32<256, true>), dim3(cus), dim3(256), TileR32::LDS_BYTES, 0, d_in, d_slots, rc, tiles, n_sc, (float2 *)nullptr, same);
                            else hipLaunchKernelGGL((tile_r32<256, false>), dim3(cus), dim3(256), TileR32::LDS_BYTES, 0, d_in, d_slots, rc, tiles, n_sc, (float2 *)nullptr, same); }, 1);
    }
    CK(hipGetLastError());
    return 0;
}
