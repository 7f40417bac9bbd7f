// xcdbench.hip -- de-risk the L2-resident hand-off: persistent workgroups form one team per XCD
// (HW_REG_XCC_ID), every iteration each workgroup writes a 64 KiB tile of its team's 2 MiB
// buffer with plain stores, the team synchronises on a device-scope counter, then every
// workgroup reads 64 KiB written by the OTHER members with sc1 (L1-bypassing) loads and checks it.
// Reports us per iteration (write + barrier + read) and the census (workgroups per XCD).
// Build: hipcc -O3 --offload-arch=gfx950 -o build/xcdbench tools/xcdbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef __attribute__((address_space(1))) unsigned gu32;

struct Ctl {
    unsigned census[8];      // workgroups registered per XCC
    unsigned arrived;        // grid-wide start counter
    unsigned pad[7];
    unsigned bar[8][16];     // per-team barrier counters (one 64 B line each)
    unsigned errors;
    unsigned timeouts;
};

__device__ __forceinline__ unsigned xcc_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

__device__ __forceinline__ unsigned ld_relaxed(unsigned *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// all threads call; leader polls
__device__ bool wait_ge(unsigned *p, unsigned target, unsigned *tmo)
{
    __shared__ int ok;
    if (threadIdx.x == 0) {
        int good = 0;
        for (unsigned spins = 0; spins < (1u << 22); spins++) {
            if (ld_relaxed(p) >= target) { good = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (!good) atomicAdd(tmo, 1u);
        ok = good;
    }
    __syncthreads();
    return ok != 0;
}

template <int MODE>   // 0: sc1 8-byte loads, 1: plain loads after an agent acquire
__global__ __launch_bounds__(512) void k_team(Ctl *ctl, float2 *mid /*[8][2 MiB]*/, int iters, unsigned *dbg)
{
    __shared__ unsigned s_xcc, s_rank, s_size;
    const int tid = threadIdx.x;
    if (tid == 0) {
        const unsigned x = xcc_id() & 7;
        s_xcc = x;
        s_rank = atomicAdd(&ctl->census[x], 1u);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        atomicAdd(&ctl->arrived, 1u);
    }
    __syncthreads();
    if (!wait_ge(&ctl->arrived, gridDim.x, &ctl->timeouts)) return;
    if (tid == 0) s_size = ld_relaxed(&ctl->census[s_xcc]);
    __syncthreads();
    const unsigned xcc = s_xcc, rank = s_rank, size = s_size;
    if (tid == 0) dbg[blockIdx.x] = xcc * 1000 + rank;
    float2 *buf = mid + (size_t)xcc * (2u << 20) / 8;           // 2 MiB per team
    const int tiles = 32;                                       // 64 KiB tiles
    unsigned bad = 0;
    for (int it = 0; it < iters; it++) {
        // write: tiles rank, rank+size, ...   (64 KiB = 8192 float2 = 512 threads x 16)
        for (unsigned t = rank; t < tiles; t += size) {
            float4 *dst = reinterpret_cast<float4 *>(buf + (size_t)t * 8192);
#pragma unroll
            for (int r = 0; r < 8; r++)
                dst[r * 512 + tid] = make_float4((float)it, (float)t, (float)(r * 512 + tid), 1.f);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) atomicAdd(&ctl->bar[xcc][0], 1u);
        if (!wait_ge(&ctl->bar[xcc][0], (2 * it + 1) * size, &ctl->timeouts)) return;
        if (MODE == 1) {
            if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        // read tiles (rank+1), ... written by other members
        for (unsigned t0 = rank; t0 < tiles; t0 += size) {
            const unsigned t = (t0 + 1) % tiles;
            const unsigned long long *src = reinterpret_cast<const unsigned long long *>(buf + (size_t)t * 8192);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                unsigned long long u;
                if (MODE == 0) u = __hip_atomic_load(src + r * 512 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else u = src[r * 512 + tid];
                // element e = r*512 + tid is float2 index; float4 index e/2, half e&1
                const unsigned e = r * 512 + tid;
                float2 f = *reinterpret_cast<float2 *>(&u);
                float2 want = (e & 1) ? make_float2((float)(e >> 1), 1.f) : make_float2((float)it, (float)t);
                if (f.x != want.x || f.y != want.y) bad++;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) atomicAdd(&ctl->bar[xcc][0], 1u);
        if (!wait_ge(&ctl->bar[xcc][0], (2 * it + 2) * size, &ctl->timeouts)) return;
    }
    if (bad) atomicAdd(&ctl->errors, bad);
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    Ctl *ctl; float2 *mid; unsigned *dbg;
    CK(hipMalloc(&ctl, sizeof(Ctl)));
    CK(hipMalloc(&mid, 16u << 20));
    CK(hipMalloc(&dbg, 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_team<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_team<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            CK(hipMemset(ctl, 0, sizeof(Ctl)));
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k_team<0>, dim3(256), dim3(512), 100 * 1024, 0, ctl, mid, iters, dbg);
            else hipLaunchKernelGGL(k_team<1>, dim3(256), dim3(512), 100 * 1024, 0, ctl, mid, iters, dbg);
            CK(hipGetLastError());
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            Ctl h; CK(hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
            printf("mode %d (%s): %.2f us/iter  census %u %u %u %u %u %u %u %u  errors %u timeouts %u\n", mode,
                   mode == 0 ? "sc1 loads" : "acquire + plain loads", ms * 1e3 / iters, h.census[0], h.census[1],
                   h.census[2], h.census[3], h.census[4], h.census[5], h.census[6], h.census[7], h.errors, h.timeouts);
        }
    }
    std::vector<unsigned> d(256);
    CK(hipMemcpy(d.data(), dbg, 256 * 4, hipMemcpyDeviceToHost));
    printf("block -> xcc*1000+rank: ");
    for (int i = 0; i < 24; i++) printf("%u ", d[i]);
    printf("\n");
    return 0;
}
