// hopbench.hip -- what does ONE notification between the workgroups of an XCD cost?  (The fused launch's hand-over of a
// half is two of them plus the row loads; with one slot per XCD that chain bounds the task period.)
// 512 one-wave workgroups; the 64 of an XCD (HW_REG_XCC_ID, arrival order) are 32 producers + 32 consumers that
// ping-pong N times: all producers signal -> every consumer has seen all 32 -> all consumers signal -> every producer
// has seen all 32.  Forms of "signal" / "see":
//   0  the launch's: every signaller adds 1 to 32 replicated counters (one wave instruction, L2 atomics of workgroup
//      scope), every waiter polls ITS replica with a scalar load (glc)
//   1  per-writer flag words: signaller r stores the round number into word r of each of the 32 waiters' lines (one wave
//      instruction, plain stores), a waiter polls its whole line (two s_load_dwordx16 glc) and takes the minimum
//   2  as 0, polled with a vector load (sc1)       3  as 1, polled with one 32-lane vector load (sc1) + DPP-free min by ballot
//   4  as 0, polled with s_dcache_inv + an ordinary scalar load (no glc)      5  as 1, polled the same way
//   6  as 5, signalled with SCALAR stores (32 s_store_dword + s_dcache_wb): the scalar path does not queue behind the
//      wave's (and the CU's) vector-memory requests
// Second argument 1: every wave issues 16 KiB of streaming loads (unwaited) in front of each signal, as a tile member does.
// Prints us per round trip (= two notifications).
// Build: hipcc -O3 --offload-arch=gfx950 -o build/tools/hopbench tools/hopbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned v16u __attribute__((ext_vector_type(16)));
struct Line { unsigned w[32]; };
struct Ctl { unsigned census[8]; unsigned bad; unsigned pad[23]; Line a[8][32], b[8][32]; unsigned long long t[8][2]; };

__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 7; }
__device__ __forceinline__ unsigned peek1(unsigned *p)
{
    unsigned v;
    asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}
__device__ __forceinline__ unsigned peek1_inv(unsigned *p)   // invalidate the scalar cache, then an ordinary (L2-served) scalar load
{
    unsigned v;
    asm volatile("s_dcache_inv\n\ts_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}
__device__ __forceinline__ unsigned peek_min32_inv(Line *p)
{
    v16u a, b;
    asm volatile("s_dcache_inv\n\ts_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a), "=&s"(b) : "s"(p) : "memory");
    unsigned m = a[0];
#pragma unroll
    for (int i = 1; i < 16; i++) m = m < a[i] ? m : a[i];
#pragma unroll
    for (int i = 0; i < 16; i++) m = m < b[i] ? m : b[i];
    return m;
}
__device__ __forceinline__ unsigned peek_min32(Line *p)
{
    v16u a, b;
    asm volatile("s_load_dwordx16 %0, %2, 0x0 glc\n\ts_load_dwordx16 %1, %2, 0x40 glc\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a), "=&s"(b) : "s"(p) : "memory");
    unsigned m = a[0];
#pragma unroll
    for (int i = 1; i < 16; i++) m = m < a[i] ? m : a[i];
#pragma unroll
    for (int i = 0; i < 16; i++) m = m < b[i] ? m : b[i];
    return m;
}
__device__ __forceinline__ void signal_scalar(Line *lines /* wave-uniform */, int rank, unsigned round)
{
    unsigned *p = &lines[0].w[rank];
#pragma unroll
    for (int i = 0; i < 32; i++) asm volatile("s_store_dword %0, %1, %2" :: "s"(round), "s"(p), "n"(i * 128) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}
template <int FORM>
__device__ __forceinline__ void signal(Line *lines, int rank, unsigned round, int l)
{
    if (FORM == 6) { signal_scalar(lines, __builtin_amdgcn_readfirstlane(rank), __builtin_amdgcn_readfirstlane(round)); return; }
    if (l < 32) {
        if (FORM == 0 || FORM == 2 || FORM == 4) __hip_atomic_fetch_add(&lines[l].w[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_store(&lines[l].w[rank], round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}
template <int FORM>
__device__ __forceinline__ bool wait_all(Line *mine, unsigned round, int l)
{
    for (unsigned spins = 0; spins < (1u << 22); spins++) {
        if (FORM == 0) { if (peek1(&mine->w[0]) >= 32u * round) return true; }
        else if (FORM == 4) { if (peek1_inv(&mine->w[0]) >= 32u * round) return true; }
        else if (FORM == 5 || FORM == 6) { if (peek_min32_inv(mine) >= round) return true; }
        else if (FORM == 1) { if (peek_min32(mine) >= round) return true; }
        else if (FORM == 2) { if (__hip_atomic_load(&mine->w[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 32u * round) return true; }
        else {
            const unsigned v = __hip_atomic_load(&mine->w[l & 31], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__builtin_amdgcn_ballot_w64(v >= round) == ~0ull) return true;
        }
    }
    return false;
}
template <int FORM>
__global__ __launch_bounds__(64) void k_hop(Ctl *ctl, int rounds, const float4 *stream, size_t stream_elems)
{
    const int l = threadIdx.x;
    const unsigned x = xcc_id();
    unsigned idx = 0;
    if (l == 0) idx = atomicAdd(&ctl->census[x], 1u);
    idx = __builtin_amdgcn_readfirstlane(idx);
    if (idx >= 64) { if (l == 0) ctl->bad = 1; return; }
    const int kind = idx & 1, rank = idx >> 1;
    // everybody of the XCD is here?
    for (unsigned spins = 0; spins < (1u << 22); spins++)
        if (__hip_atomic_load(&ctl->census[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 64u) break;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    bool ok = true;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = 1; r <= rounds && ok; r++) {
        if (stream) {   // 16 KiB of streaming loads per wave in front of the signal, consumed a round later
            const size_t base = (((size_t)blockIdx.x * rounds + r) * 1024) % (stream_elems - 1024);
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const float4 v = stream[base + j * 64 + l];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
        if (kind == 0) {
            signal<FORM>(ctl->a[x], rank, r, l);
            ok = wait_all<FORM>(&ctl->b[x][rank], r, l);
        } else {
            ok = wait_all<FORM>(&ctl->a[x][rank], r, l);
            signal<FORM>(ctl->b[x], rank, r, l);
        }
    }
    if (!ok && l == 0) ctl->bad = 2;
    if (acc.x + acc.y + acc.z + acc.w == 12345.f) ctl->bad = 3;
    if (idx == 0 && l == 0) { ctl->t[x][0] = t0; ctl->t[x][1] = __builtin_amdgcn_s_memrealtime(); }
}
int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
    const int loaded = argc > 2 ? atoi(argv[2]) : 0;
    Ctl *d, h;
    CK(hipMalloc(&d, sizeof(Ctl)));
    float4 *stream = nullptr;
    const size_t stream_elems = (size_t)1 << 26;   // 1 GiB
    if (loaded) { CK(hipMalloc(&stream, stream_elems * sizeof(float4))); CK(hipMemset(stream, 0, stream_elems * sizeof(float4))); }
    printf("%s\n", loaded ? "every wave streams 16 KiB in front of each signal" : "idle chip");
#define RUN(F) do { CK(hipMemset(d, 0, sizeof(Ctl))); hipLaunchKernelGGL(k_hop<F>, dim3(512), dim3(64), 0, 0, d, rounds, stream, stream_elems); CK(hipDeviceSynchronize()); \
        CK(hipMemcpy(&h, d, sizeof(Ctl), hipMemcpyDeviceToHost)); double s = 0; int n = 0; \
        for (int x = 0; x < 8; x++) if (h.t[x][1]) { s += (h.t[x][1] - h.t[x][0]) / 100.0 / rounds; n++; } \
        printf("form %d: %.3f us per round trip (%d teams, bad %u)\n", F, n ? s / n : 0.0, n, h.bad); } while (0)
    RUN(0); RUN(1); RUN(2); RUN(3); RUN(4); RUN(5); RUN(6); RUN(5); RUN(6);
    return 0;
}
