// rolebench.hip -- who does what on a CU?  The work of the fused 1024 x 512 launch WITHOUT its hand-over protocol (every
// member runs free: no flags, no looks; the rows transform whatever lies in the slot), two ways of dealing it to the two
// workgroups a CU holds:
//
//   S  (split, the launch's structure): one TILE workgroup (a range tile per task, input from HBM) beside one ROW workgroup
//      (sixteen Doppler rows per task, loaded from the L2-resident slot).  The tile workgroup's serial chain -- stage 1, four
//      barriers, two LDS-bound stage-2/3 phases -- is the pace of the launch; the row waves fill its gaps and poll.
//   U  (uniform): BOTH workgroups alternate -- a tile, then two row jobs (one row per wave each), then the next tile.  Per
//      CU and task the same work; each workgroup's chain is (tile + 2 row jobs) per TWO tasks, and the workgroup in its row
//      phase keeps its next tile's requests in flight (64 landing registers per workgroup = twice the landing capacity).
//      The row jobs run in the registers the landing zone leaves (Doppler twiddles read per row), their wave buffers
//      overlay the tile image, whose twiddle pads are restored in front of each tile.
//
// Reported: us per task and CU (a task = one tile + sixteen rows), sector-equivalent (8 teams, two tasks per sector).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -I weather-radar-processing_amd/csrc
//              -o build/tools/rolebench tools/rolebench.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "wrp_fused.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

namespace wrp {

struct RoleCtl {
    unsigned arrivals[8][256];   // workgroups seen per CU (xcc, hw key)
    unsigned tile_done[8][256];  // mode S: tasks the CU's tile workgroup has finished (the row workgroup follows it: two rows per wave and task)
    unsigned long long rows_done;
};

__device__ __forceinline__ void tile_tables(unsigned char *smem, const RangeConsts &rc, int tid)
{
    typedef FusedTile T;
    for (int e = tid; e < RP_M; e += FUSED_THREADS) {
        const int p0 = e >> 4, k1 = e & 15;
        *reinterpret_cast<float2 *>(smem + T::tw1_addr(p0, k1)) = rc.tw[(p0 * k1) & (RP_M - 1)];
        reinterpret_cast<float *>(smem + T::OFF_WR)[e] = rc.wr_c[e];
    }
    if (tid < 64) *reinterpret_cast<float2 *>(smem + T::tw2_addr(tid >> 3, tid & 7)) = rc.tw[(16 * (tid >> 3) * (tid & 7)) & (RP_M - 1)];
}

// one tile task of the launch (tile member's loop body, no looks / flags); the next tile is requested on the way
template <bool LOAD>
__device__ __forceinline__ void tile_task(unsigned char *smem, float4 (&v)[16], float2 &wdv, const RangeConsts &rc, float2 *mid, int store_col,
                                          const float2 *next, int next_col, bool more)
{
    cf ga[8], gc[8];
    FusedStage1Tables s1t;
    fused_stage1_tables(smem, s1t);
    fused_stage1<0>(smem, v, wdv, s1t, ga);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    fused_stage1<1>(smem, v, wdv, s1t, gc);
    __syncthreads();                    // A1
    cf o[2][4];
    const int voff_next = fused_tile_voff(next_col);
#define RB_L1(R) fused_tile_load1<R>(next, voff_next, rc.wd, v, wdv, LOAD && more)
    RB_L1(0); RB_L1(8);
    fused_stage2_item<0>(smem);
    RB_L1(4);
    fused_stage2_item<1>(smem);
    RB_L1(12);
    fused_stage3_item<0>(smem, o);
    RB_L1(1);
    fused_stage3_item<1>(smem, o);
    RB_L1(9);
    __syncthreads();                    // A2
    fused_store(mid, store_col, 0, o);
    __builtin_amdgcn_sched_barrier(0);
    RB_L1(5); RB_L1(13); RB_L1(2); RB_L1(10);
    fused_group1_to_lds(smem, ga, gc);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __syncthreads();                    // A3
    RB_L1(6);
    fused_stage2_item<0>(smem);
    RB_L1(14);
    fused_stage2_item<1>(smem);
    RB_L1(3);
    fused_stage3_item<0>(smem, o);
    RB_L1(11);
    fused_stage3_item<1>(smem, o);
    RB_L1(7); RB_L1(15);
#undef RB_L1
    __syncthreads();                    // A4
    fused_store(mid, store_col, 1, o);
}

template <int MODE, bool LOAD>   // MODE 0: S, 1: U
__global__ __launch_bounds__(FUSED_THREADS, 4) __attribute__((amdgpu_waves_per_eu(4, 4))) void role_kernel(
    const float2 *__restrict__ iq, float2 *slots, float *out, RoleCtl *ctl, RangeConsts rc, const float2 *__restrict__ tw_n, MaTaps taps,
    int tasks, int n_sc)
{
    typedef FusedTile T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_word *s_ctl = (lds_word *)(smem + T::OFF_CTL);
    const int tid = threadIdx.x, w = wave_id(), l = tid & 63;
    if (tid == 0) {
        const unsigned x = xcc_id(), key = hw_cu_key();
        s_ctl[0] = (int)__hip_atomic_fetch_add(&ctl->arrivals[x][key], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_ctl[1] = (int)x;
        s_ctl[2] = (int)key;
    }
    __syncthreads();
    const int kind = __builtin_amdgcn_readfirstlane(s_ctl[0]) & 1, xcc = __builtin_amdgcn_readfirstlane(s_ctl[1]),
              key = __builtin_amdgcn_readfirstlane(s_ctl[2]);
    // input tiles as in the launch: the workgroups of an XCD read one matrix per task, 32 column tiles rotated
    const int member = (blockIdx.x >> 3) & 31;
    auto tile_src = [&](int t) { return iq + (size_t)((xcc + 8 * t) % n_sc) * RP_M * DP_N; };
    auto tile_col = [&](int t) { return ((member + t) & 31) * 16; };
    float2 *mid = slots + (size_t)xcc * FUSED_TEAM_ELEMS;
    const int store_col = member * 16;
    float2 *wbuf = reinterpret_cast<float2 *>(smem) + (size_t)w * DP_ELEMS;
    float2 *s_twn = reinterpret_cast<float2 *>(smem + T::OFF_TWN);
    const DumpPtrs nodump{};
    float acc = 0.f;

    if (MODE == 0) {
        if (kind == 0) {
            float4 v[16];
            float2 wdv;
            fused_tile_load<0>(tile_src(0), tile_col(0), rc.wd, v, wdv, LOAD);
            fused_tile_load<1>(tile_src(0), tile_col(0), rc.wd, v, wdv, LOAD);
            fused_tile_load<2>(tile_src(0), tile_col(0), rc.wd, v, wdv, LOAD);
            fused_tile_load<3>(tile_src(0), tile_col(0), rc.wd, v, wdv, LOAD);
            tile_tables(smem, rc, tid);
            __syncthreads();
#pragma unroll 1
            for (int t = 0; t < tasks; t++) {
                tile_task<LOAD>(smem, v, wdv, rc, mid, store_col, tile_src(t + 1 < tasks ? t + 1 : 0), tile_col(t + 1), t + 1 < tasks);
                if (tid == 0) __hip_atomic_store(&ctl->tile_done[xcc][key], (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            doppler_twiddles_to_lds(s_twn, tw_n, tid, FUSED_THREADS);
            __syncthreads();
            DopplerTwiddles row_tw;
            doppler_row_twiddles(s_twn, l, row_tw);
            unsigned long long rows = 0;
#pragma unroll 1
            for (int t = 0; t < tasks; t++) {
                // the rows of task t follow the CU's own tile task t (in the launch: the team's; two rows per wave and task)
                while ((int)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&ctl->tile_done[xcc][key], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < t)
                    __builtin_amdgcn_s_sleep(2);
#pragma unroll 1
                for (int g = 0; g < 2; g++) {
                    cf x[8];
                    doppler_load_row<AUX_SC1>(mid + (size_t)((member * 8 + w + 7 * t + 128 * g) & 255) * DP_N, l, x);
                    acc += doppler_row<false, 7, true>(x, wbuf, s_twn, taps, l, w, false, nodump, row_tw);
                    rows++;
                }
            }
            if (l == 0) atomicAdd(&ctl->rows_done, rows);
        }
    } else {
        // uniform: tile, two row jobs, tile, ...  `kind` only staggers the two workgroups of a CU by half a cycle
        float4 v[16];
        float2 wdv;
        fused_tile_load<0>(tile_src(kind), tile_col(kind), rc.wd, v, wdv, LOAD);
        fused_tile_load<1>(tile_src(kind), tile_col(kind), rc.wd, v, wdv, LOAD);
        fused_tile_load<2>(tile_src(kind), tile_col(kind), rc.wd, v, wdv, LOAD);
        fused_tile_load<3>(tile_src(kind), tile_col(kind), rc.wd, v, wdv, LOAD);
        auto row_job = [&](int t) {
            cf x[8];
            doppler_load_row<AUX_SC1>(mid + (size_t)((member * 8 + w + 7 * t) & 255) * DP_N, l, x);
            acc += doppler_row<false, 7, false>(x, wbuf, s_twn, taps, l, w, false, nodump);
        };
        if (kind) {   // the second workgroup of a CU starts with its row jobs
            doppler_twiddles_to_lds(s_twn, tw_n, tid, FUSED_THREADS);
            __syncthreads();
            row_job(0);
            row_job(1);
            __syncthreads();
        }
#pragma unroll 1
        for (int t = kind; t < tasks; t += 2) {
            tile_tables(smem, rc, tid);          // (the launch would restore only the pads the row buffers overlay)
            __syncthreads();
            tile_task<LOAD>(smem, v, wdv, rc, mid, store_col, tile_src(t + 2 < tasks ? t + 2 : 0), tile_col(t + 2), t + 2 < tasks);
            __syncthreads();                     // stage 3 of every wave has read the image
            doppler_twiddles_to_lds(s_twn, tw_n, tid, FUSED_THREADS);
            __syncthreads();
            row_job(t);
            row_job(t + 1);
            __syncthreads();                     // the wave buffers are free: the image may be written
        }
    }
    if (acc == 123.456f) out[blockIdx.x] = acc;
}

} // namespace wrp

int main(int argc, char **argv)
{
    using namespace wrp;
    int tasks = 128;
    for (int a = 1; a < argc; a++)
        if (!strcmp(argv[a], "--tasks") && a + 1 < argc) tasks = atoi(argv[++a]);
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int n_sc = 128;
    const size_t sc_elems = (size_t)RP_M * DP_N;
    std::vector<float2> h_in(sc_elems);
    srand(5);
    for (auto &x : h_in) x = make_float2((float)(rand() % 32768 - 16384), (float)(rand() % 32768 - 16384));
    float2 *d_in, *d_slots;
    float *d_out;
    RoleCtl *d_ctl;
    CK(hipMalloc(&d_in, sizeof(float2) * sc_elems * n_sc));
    for (int s = 0; s < n_sc; s++) CK(hipMemcpy(d_in + s * sc_elems, h_in.data(), sizeof(float2) * sc_elems, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_slots, sizeof(float2) * FUSED_TEAM_ELEMS * 8));
    CK(hipMemset(d_slots, 0, sizeof(float2) * FUSED_TEAM_ELEMS * 8));
    CK(hipMalloc(&d_out, 4 * 4096));
    CK(hipMalloc(&d_ctl, sizeof(RoleCtl)));
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
    std::vector<float2> tw(RP_M), twn(DP_TW_ELEMS);
    for (int k = 0; k < RP_M; k++) tw[k] = make_float2((float)std::cos(2 * M_PI * k / RP_M), (float)-std::sin(2 * M_PI * k / RP_M));
    for (int e = 0; e < DP_TW_ELEMS; e++) {
        const int k = doppler_twiddle_index(e);
        twn[e] = make_float2((float)std::cos(2 * M_PI * k / DP_N), (float)std::sin(2 * M_PI * k / DP_N));
    }
    float *d_wr, *d_wd;
    float2 *d_tw, *d_twn;
    CK(hipMalloc(&d_wr, 4 * RP_M)); CK(hipMalloc(&d_wd, 4 * DP_N)); CK(hipMalloc(&d_tw, 8 * RP_M)); CK(hipMalloc(&d_twn, 8 * DP_TW_ELEMS));
    CK(hipMemcpy(d_wr, wr.data(), 4 * RP_M, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_wd, wd.data(), 4 * DP_N, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tw, tw.data(), 8 * RP_M, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_twn, twn.data(), 8 * DP_TW_ELEMS, hipMemcpyHostToDevice));
    const RangeConsts rc{d_wr, d_wd, d_tw};
    MaTaps taps;
    {
        double g[9], sum = 0;
        for (int i = 0; i < 7; i++) { g[i] = std::exp(-std::pow(i - 3, 2.0) / 2); sum += g[i]; }
        for (int i = 0; i < 9; i++) taps.g[i] = i < 7 ? (float)(g[i] / sum) : 0.f;
    }
#define ATTR(M, L) CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&role_kernel<M, L>), hipFuncAttributeMaxDynamicSharedMemorySize, FusedTile::LDS_BYTES))
    ATTR(0, true); ATTR(0, false); ATTR(1, true); ATTR(1, false);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int load = 1; load >= 0; load--)
        for (int mode = 0; mode < 2; mode++) {
            float best = 1e30f;
            unsigned long long rows = 0;
            for (int rep = 0; rep < 4; rep++) {
                CK(hipMemset(d_ctl, 0, sizeof(RoleCtl)));
                CK(hipEventRecord(e0));
#define GO(M, L) hipLaunchKernelGGL((role_kernel<M, L>), dim3(2 * cus), dim3(FUSED_THREADS), FusedTile::LDS_BYTES, 0, d_in, d_slots, d_out, d_ctl, rc, d_twn, taps, tasks, n_sc)
                if (mode == 0) { if (load) GO(0, true); else GO(0, false); }
                else { if (load) GO(1, true); else GO(1, false); }
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best) {
                    best = ms;
                    RoleCtl h;
                    CK(hipMemcpy(&h, d_ctl, sizeof h, hipMemcpyDeviceToHost));
                    rows = h.rows_done;
                }
            }
            const double us_task = best * 1e3 / tasks;
            printf("%s, %-14s %7.3f us per task and CU, sector-equivalent %6.3f us", mode ? "U (uniform workgroups)   " : "S (tile + row workgroup) ",
                   load ? "input from HBM" : "input dropped", us_task, us_task / 4.0);
            if (mode == 0) printf("   (row workgroups transformed %.1f rows per task and CU; the launch needs 16)", (double)rows / ((double)tasks * cus));
            printf("\n");
        }
    CK(hipGetLastError());
    return 0;
}
