// phasebench.hip -- the two item kinds of the fused launch (wrp_fused.h) in isolation: no counters,
// no team, every workgroup (1024 threads, one per CU) repeats its own item K times.
//   A: tile load (next tile prefetched) -> stages 1-2 -> stage 3 store into a private mid region
//   B: 16 rows (sc1 loads from a private, L2-resident region) -> doppler_row -> one store per row
//      optionally with a tile prefetch in flight (as the fused launch does)
// Reports us per item per CU; the fused launch cannot be faster than A + B per item pair.
// Build: hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -Iweather-radar-processing_amd/csrc -o build/phasebench tools/phasebench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "wrp_fused.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
using namespace wrp;
typedef FusedGeom<16> G;
typedef G::FT FT;
constexpr int FUSED_THREADS = G::THREADS, FUSED_LDS_BYTES = G::LDS_BYTES, FUSED_OFF_TWN = G::OFF_TWN;
#define fused_tile_load fused_tile_load<16>
#define fused_stage12 fused_stage12<16>
#define fused_stage3_compute fused_stage3_compute<16>
#define fused_stage3_store fused_stage3_store<16>

__device__ __forceinline__ void tables(unsigned char *smem, RangeConsts rc, const float2 *tw_n)
{
    const int tid = threadIdx.x;
    float2 *s_twn = reinterpret_cast<float2 *>(smem + FUSED_OFF_TWN);
    if (tid < DP_N) s_twn[tid] = tw_n[tid];
    *reinterpret_cast<float2 *>(smem + FT::tw_addr(tid)) = rc.tw[tid];
    reinterpret_cast<float *>(smem + FT::OFF_WR)[tid] = rc.wr_c[tid];
    __syncthreads();
}

// MODE 0: A items; 1: B items; 2: B items with a tile prefetch in flight; 3: A then B alternating;
// 4 / 5: as 2 / 3 with every prefetch taken from a different channel of a 2 GiB input (HBM misses, as in the real launch)
constexpr int NCH = 512;
template <int MODE>
__global__ __launch_bounds__(FUSED_THREADS) void k_phase(const float2 *iq, float2 *mid, float *out, RangeConsts rc,
                                                          const float2 *tw_n, MaTaps taps, int K)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    tables(smem, rc, tw_n);
    const int tid = threadIdx.x, w = wave_id(), l = tid & 63, n = DP_N;
    float2 *s_twn = reinterpret_cast<float2 *>(smem + FUSED_OFF_TWN);
    float2 *wbuf = reinterpret_cast<float2 *>(smem) + (size_t)w * DP_ELEMS;
    const DumpPtrs nodump{};
    // private regions: a [1024][512] channel of input per 32 workgroups (tile = blockIdx % 32), 16 rows of mid per workgroup
    const float2 *src = iq + (size_t)(blockIdx.x / 32) * RP_M * n;
    const int col = (blockIdx.x % 32) * 16;
    float2 *mymid = mid + (size_t)(blockIdx.x / 32) * FUSED_MID_ELEMS;   // whole channel; this workgroup's tile columns
    float2 *myrows = mid + (size_t)(blockIdx.x / 32) * FUSED_MID_ELEMS + (size_t)(blockIdx.x % 32) * 16 * n;
    float4 v[8];
    float2 wdv;
    if (MODE != 1) fused_tile_load(src, n, col, rc.wd, v, wdv, true);
#pragma unroll 1
    for (int k = 0; k < K; k++) {
        if (MODE == 0 || MODE == 3 || MODE == 5) {
            const float2 wcur = wdv;
            fused_stage12(smem, v, wcur, [] {}, [&]() { if (MODE == 0) fused_tile_load(src, n, col, rc.wd, v, wdv, true); });
            float4 o[4];
            fused_stage3_compute(smem, o);
            fused_stage3_store(mymid, n, col, o);
            __syncthreads();
        }
        if (MODE >= 1) {
            cf x[8];
            doppler_load_row<AUX_SC1>(myrows + (size_t)w * n, l, x);
            if (MODE >= 2) fused_tile_load(MODE >= 4 ? iq + (size_t)((blockIdx.x / 32 + 8 * (k + 1)) % NCH) * RP_M * n : src, n, col, rc.wd, v, wdv, true);
            const float s = doppler_row<false, 7>(x, wbuf, s_twn, taps, l, w, false, nodump);
            if (l == 0) out[blockIdx.x * 16 + w] = s;
            __syncthreads();
        }
    }
    if (MODE != 1) {   // keep the last prefetch alive
        float a = 0;
        for (int r = 0; r < 8; r++) a += v[r].x;
        if (a == 12345.f) out[0] = a + wdv.x;
    }
}

template <int MODE>
void run(const float2 *iq, float2 *mid, float *out, RangeConsts rc, const float2 *tw_n, const char *label)
{
    const int K = 200;
    MaTaps taps{};
    for (int i = 0; i < 7; i++) taps.g[i] = 1.f / 7;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_phase<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, FUSED_LDS_BYTES));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_phase<MODE>, dim3(256), dim3(FUSED_THREADS), FUSED_LDS_BYTES, 0, iq, mid, out, rc, tw_n, taps, K);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("%-44s %7.2f us per iteration per CU\n", label, ms * 1e3 / K);
}

int main()
{
    float2 *iq, *mid, *tw_m, *tw_n; float *out, *wr, *wd;
    CK(hipMalloc(&iq, (size_t)NCH * RP_M * DP_N * 8)); CK(hipMemset(iq, 0, (size_t)NCH * RP_M * DP_N * 8)); CK(hipMalloc(&mid, 8ull * FUSED_MID_ELEMS * 8));
    CK(hipMalloc(&tw_m, 1024 * 8)); CK(hipMalloc(&tw_n, 512 * 8)); CK(hipMalloc(&out, 1 << 20));
    CK(hipMalloc(&wr, 1024 * 4)); CK(hipMalloc(&wd, 512 * 4));
    std::vector<float2> h(8ull * RP_M * DP_N);
    for (size_t i = 0; i < h.size(); i++) h[i] = make_float2((float)(i % 977) * 1e-3f, (float)(i % 331) * -2e-3f);
    CK(hipMemcpy(iq, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(mid, h.data(), 8ull * FUSED_MID_ELEMS * 8, hipMemcpyHostToDevice));
    std::vector<float2> t(1024);
    for (int k = 0; k < 1024; k++) t[k] = make_float2(cosf(6.2831853f * k / 1024), -sinf(6.2831853f * k / 1024));
    CK(hipMemcpy(tw_m, t.data(), 1024 * 8, hipMemcpyHostToDevice));
    for (int k = 0; k < 512; k++) t[k] = make_float2(cosf(6.2831853f * k / 512), sinf(6.2831853f * k / 512));
    CK(hipMemcpy(tw_n, t.data(), 512 * 8, hipMemcpyHostToDevice));
    std::vector<float> f(1024, 0.5f);
    CK(hipMemcpy(wr, f.data(), 1024 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(wd, f.data(), 512 * 4, hipMemcpyHostToDevice));
    const RangeConsts rc{wr, wd, tw_m};
    run<0>(iq, mid, out, rc, tw_n, "A: tile -> stages 1-2 -> stage 3 store");
    run<1>(iq, mid, out, rc, tw_n, "B: 16 rows, no prefetch");
    run<2>(iq, mid, out, rc, tw_n, "B: 16 rows, tile prefetch in flight");
    run<3>(iq, mid, out, rc, tw_n, "A then B (prefetch during B)");
    run<4>(iq, mid, out, rc, tw_n, "B: 16 rows, HBM tile prefetch in flight");
    run<5>(iq, mid, out, rc, tw_n, "A then B (HBM prefetch during B)");
    return 0;
}
