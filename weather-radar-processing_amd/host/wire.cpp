// wire.cpp -- see wire.h.
#include "wire.h"

#include <stdint.h>
#include <string.h>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

void wire_drop_vh_portable(unsigned char *dst8, const unsigned char *src12, size_t samples)
{
    for (size_t i = 0; i < samples; i++) memcpy(dst8 + 8 * i, src12 + 12 * i, 8);
}

#if defined(__x86_64__)
// four samples (48 bytes) -> 32 bytes with byte shuffles: out0 = A[0..7] A[12..15] B[0..3], out1 = B[8..15] C[4..11].
// STREAM: non-temporal stores (the destination is 16-byte aligned).  The destination is a pinned slot that the GPU's DMA
// engine reads next and the CPU never reads again: written through the caches it would sit there dirty, and every line of
// the H2D copy would have to be snooped out of them (feeder_breakdown.log: with cached stores the transfer behind a copying
// feeder ran at 96 us per 4 MiB sector against 75 us behind an idle one).
template <bool STREAM>
__attribute__((target("ssse3"))) static void drop_vh_ssse3(unsigned char *dst8, const unsigned char *src12, size_t samples)
{
    const __m128i a0 = _mm_setr_epi8(0, 1, 2, 3, 4, 5, 6, 7, 12, 13, 14, 15, -1, -1, -1, -1);
    const __m128i b0 = _mm_setr_epi8(-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 0, 1, 2, 3);
    const __m128i b1 = _mm_setr_epi8(8, 9, 10, 11, 12, 13, 14, 15, -1, -1, -1, -1, -1, -1, -1, -1);
    const __m128i c1 = _mm_setr_epi8(-1, -1, -1, -1, -1, -1, -1, -1, 4, 5, 6, 7, 8, 9, 10, 11);
    size_t i = 0;
    for (; i + 4 <= samples; i += 4) {
        const __m128i A = _mm_loadu_si128((const __m128i *)(src12 + 12 * i));
        const __m128i B = _mm_loadu_si128((const __m128i *)(src12 + 12 * i + 16));
        const __m128i C = _mm_loadu_si128((const __m128i *)(src12 + 12 * i + 32));
        const __m128i o0 = _mm_or_si128(_mm_shuffle_epi8(A, a0), _mm_shuffle_epi8(B, b0));
        const __m128i o1 = _mm_or_si128(_mm_shuffle_epi8(B, b1), _mm_shuffle_epi8(C, c1));
        if (STREAM) {
            _mm_stream_si128((__m128i *)(dst8 + 8 * i), o0);
            _mm_stream_si128((__m128i *)(dst8 + 8 * i + 16), o1);
        } else {
            _mm_storeu_si128((__m128i *)(dst8 + 8 * i), o0);
            _mm_storeu_si128((__m128i *)(dst8 + 8 * i + 16), o1);
        }
    }
    if (STREAM) _mm_sfence();
    wire_drop_vh_portable(dst8 + 8 * i, src12 + 12 * i, samples - i);
}

// plain copy with non-temporal stores (dst 16-byte aligned, bytes a multiple of 64): the 12-byte feeder's copy into a pinned slot
static void copy_stream(char *dst, const char *src, size_t bytes)
{
    for (size_t o = 0; o < bytes; o += 64) {
        const __m128i a = _mm_loadu_si128((const __m128i *)(src + o)), b = _mm_loadu_si128((const __m128i *)(src + o + 16));
        const __m128i c = _mm_loadu_si128((const __m128i *)(src + o + 32)), d = _mm_loadu_si128((const __m128i *)(src + o + 48));
        _mm_stream_si128((__m128i *)(dst + o), a);
        _mm_stream_si128((__m128i *)(dst + o + 16), b);
        _mm_stream_si128((__m128i *)(dst + o + 32), c);
        _mm_stream_si128((__m128i *)(dst + o + 48), d);
    }
    _mm_sfence();
}
#endif

void wire_copy_to_pinned(char *dst, const char *src, size_t bytes)
{
#if defined(__x86_64__)
    if (((uintptr_t)dst & 15) == 0) {
        const size_t body = bytes & ~(size_t)63;
        copy_stream(dst, src, body);
        memcpy(dst + body, src + body, bytes - body);
        return;
    }
#endif
    memcpy(dst, src, bytes);
}

void wire_drop_vh(unsigned char *dst8, const unsigned char *src12, size_t samples)
{
#if defined(__x86_64__)
    static const bool ssse3 = __builtin_cpu_supports("ssse3");
    if (ssse3) {
        if (((uintptr_t)dst8 & 15) == 0) drop_vh_ssse3<true>(dst8, src12, samples);
        else drop_vh_ssse3<false>(dst8, src12, samples);
        return;
    }
#endif
    wire_drop_vh_portable(dst8, src12, samples);
}

// ---- FillPool -----------------------------------------------------------------------------------------------------------
FillPool::FillPool(int threads) : n_(threads < 1 ? 1 : threads)
{
    for (int t = 1; t < n_; t++) workers_.emplace_back([this, t] { worker(t); });
}

FillPool::~FillPool()
{
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; gen_.fetch_add(1); }
    cv_.notify_all();
    for (auto &w : workers_) w.join();
}

void FillPool::part(int t)
{
    // thread t's share: whole 4 KiB pages of the destination (copy) / whole groups of 512 samples (drop_vh)
    const size_t grain = drop_ ? 512 : 4096;
    const size_t chunk = ((units_ + n_ - 1) / n_ + grain - 1) / grain * grain, lo = (size_t)t * chunk;
    if (lo >= units_) return;
    const size_t cnt = lo + chunk <= units_ ? chunk : units_ - lo;
    if (drop_) wire_drop_vh((unsigned char *)dst_ + 8 * lo, (const unsigned char *)src_ + 12 * lo, cnt);
    else wire_copy_to_pinned(dst_ + lo, src_ + lo, cnt);
}

void FillPool::run_job()
{
    if (n_ == 1) { part(0); return; }
    left_.store(n_ - 1, std::memory_order_relaxed);
    gen_.fetch_add(1, std::memory_order_release);            // publishes dst_ / src_ / units_ / drop_ with it
    if (asleep_.load(std::memory_order_acquire) > 0) { std::lock_guard<std::mutex> lk(mu_); cv_.notify_all(); }
    part(0);
    while (left_.load(std::memory_order_acquire) != 0) {
#if defined(__x86_64__)
        _mm_pause();
#endif
    }
}

void FillPool::copy(char *dst, const char *src, size_t bytes)
{
    dst_ = dst; src_ = src; units_ = bytes; drop_ = false;
    run_job();
}

void FillPool::drop_vh(char *dst8, const char *src12, size_t samples)
{
    dst_ = dst8; src_ = src12; units_ = samples; drop_ = true;
    run_job();
}

void FillPool::worker(int t)
{
    long seen = 0;
    for (;;) {
        // the next job is usually less than a sector's time away: spin for it (~50 us), then sleep
        int spins = 0;
        while (gen_.load(std::memory_order_acquire) == seen) {
            if (++spins < 20000) {
#if defined(__x86_64__)
                _mm_pause();
#endif
                continue;
            }
            std::unique_lock<std::mutex> lk(mu_);
            asleep_.fetch_add(1, std::memory_order_release);
            cv_.wait(lk, [&] { return gen_.load(std::memory_order_acquire) != seen; });
            asleep_.fetch_sub(1, std::memory_order_release);
        }
        seen = gen_.load(std::memory_order_acquire);
        { std::lock_guard<std::mutex> lk(mu_); if (stop_) return; }
        part(t);
        left_.fetch_sub(1, std::memory_order_release);
    }
}
