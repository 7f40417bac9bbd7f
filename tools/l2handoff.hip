// l2handoff.hip -- how does a CU read data another CU of the SAME XCD has just written, out of that
// XCD's L2 and not out of its own (stale) L1 or over the fabric?  One 512-thread workgroup per CU,
// teams by HW_REG_XCC_ID; every iteration each member overwrites its 64 KiB tile of the team's
// 2 MiB buffer (plain stores), the team meets (L2 atomics), then every member reads and checks the
// 8 tiles behind its own (512 KiB, written by other CUs in THIS iteration) with one of:
//   0 device-scope loads (sc1)            3 nontemporal loads (nt)
//   1 buffer_inv sc1, then plain loads    4 sc0 loads
//   2 buffer_inv sc0, then plain loads    5 sc0 nt loads          6 plain loads (expected: stale)
// Reports us per iteration and the number of stale values seen.  Fast and 0 errors = served by the L2.
// Build: hipcc -O3 --offload-arch=gfx950 -o build/l2handoff tools/l2handoff.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Ctl {
    unsigned census[8];
    unsigned arrived, errors, timeouts, pad[5];
    unsigned bar[8][16];
};
typedef unsigned v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned xcc_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7;
}
__device__ __forceinline__ unsigned l2_peek(unsigned *p)
{
    unsigned zero = 0;
    asm volatile("" : "+v"(zero));
    return __hip_atomic_fetch_add(p, zero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <bool TEAM>
__device__ bool wait_ge(unsigned *p, unsigned target, unsigned *tmo)
{
    __shared__ int ok;
    if (threadIdx.x == 0) {
        int good = 0;
        for (unsigned spins = 0; spins < (1u << 20); spins++) {
            const unsigned now = TEAM ? l2_peek(p) : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (now >= target) { good = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (!good) atomicAdd(tmo, 1u);
        ok = good;
    }
    __syncthreads();
    const bool r = ok != 0;
    __syncthreads();
    return r;
}

template <int MODE>
__global__ __launch_bounds__(512) void k_team(Ctl *ctl, float2 *mid /*[8][2 MiB]*/, int iters)
{
    __shared__ unsigned s_xcc, s_rank, s_size;
    const int tid = threadIdx.x;
    if (tid == 0) {
        const unsigned x = xcc_id();
        s_xcc = x;
        s_rank = atomicAdd(&ctl->census[x], 1u);
        __hip_atomic_fetch_add(&ctl->arrived, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!wait_ge<false>(&ctl->arrived, gridDim.x, &ctl->timeouts)) return;
    if (tid == 0) s_size = __hip_atomic_load(&ctl->census[s_xcc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned xcc = s_xcc, rank = s_rank, size = s_size;
    if (size != 32 || rank >= 32) return;
    float2 *buf = mid + (size_t)xcc * (2u << 20) / 8;           // 2 MiB per team = 32 tiles of 64 KiB
    unsigned *bar = &ctl->bar[xcc][0];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 2 << 20, 0x00020000);
    unsigned bad = 0;
    for (int it = 0; it < iters; it++) {
        float4 *dst = reinterpret_cast<float4 *>(buf + (size_t)rank * 8192);
#pragma unroll
        for (int r = 0; r < 8; r++) dst[r * 512 + tid] = make_float4((float)it, (float)rank, (float)(r * 512 + tid), 1.f);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (!wait_ge<true>(bar, (2 * it + 1) * 32, &ctl->timeouts)) return;
        if (MODE == 1) asm volatile("buffer_inv sc1" ::: "memory");
        if (MODE == 2) asm volatile("buffer_inv sc0" ::: "memory");
#pragma unroll 1
        for (unsigned d = 1; d <= 8; d++) {
            const unsigned t = (rank + d) % 32;
            const unsigned long long *src = reinterpret_cast<const unsigned long long *>(buf + (size_t)t * 8192);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const unsigned e = r * 512 + tid;
                unsigned long long u;
                if (MODE == 0) u = __hip_atomic_load(src + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else if (MODE == 3) u = __builtin_nontemporal_load(src + e);
                else if (MODE == 4) u = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)((t * 8192 + e) * 8), 0, 1));
                else if (MODE == 5) u = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)((t * 8192 + e) * 8), 0, 3));
                else u = src[e];
                const float2 f = __builtin_bit_cast(float2, u);
                const float2 want = (e & 1) ? make_float2((float)(e >> 1), 1.f) : make_float2((float)it, (float)t);
                if (f.x != want.x || f.y != want.y) bad++;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (!wait_ge<true>(bar, (2 * it + 2) * 32, &ctl->timeouts)) return;
    }
    if (bad) atomicAdd(&ctl->errors, bad);
}

template <int MODE>
void run(Ctl *ctl, float2 *mid, int iters, const char *label)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    Ctl h;
    for (int rep = 0; rep < 2; rep++) {
        CK(hipMemset(ctl, 0, sizeof(Ctl)));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_team<MODE>, dim3(256), dim3(512), 100 * 1024, 0, ctl, mid, iters);
        CK(hipGetLastError());
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
    }
    printf("%-34s %8.2f us/iter  stale %10u  timeouts %u  (census %u..)\n", label, ms * 1e3 / iters, h.errors, h.timeouts, h.census[0]);
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    Ctl *ctl; float2 *mid;
    CK(hipMalloc(&ctl, sizeof(Ctl)));
    CK(hipMalloc(&mid, 16u << 20));
#define ATTR(M) CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_team<M>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024))
    ATTR(0); ATTR(1); ATTR(2); ATTR(3); ATTR(4); ATTR(5); ATTR(6);
    run<6>(ctl, mid, iters, "6 plain loads");
    run<0>(ctl, mid, iters, "0 device-scope loads (sc1)");
    run<1>(ctl, mid, iters, "1 buffer_inv sc1 + plain loads");
    run<2>(ctl, mid, iters, "2 buffer_inv sc0 + plain loads");
    run<3>(ctl, mid, iters, "3 nontemporal loads (nt)");
    run<4>(ctl, mid, iters, "4 sc0 loads");
    run<5>(ctl, mid, iters, "5 sc0 nt loads");
    return 0;
}
