// wire.h -- the feeder's side of WRP_FLAG_WIRE_8 (include/wrp.h).
//
// A sector arrives as 12 bytes per sample: hhI hhQ vvI vvQ vhI vhQ, big-endian int16 (sector.cpp:52-62).  No output reads
// VH (rpv2.cu:199-213), and the feeder copies every byte once anyway (socket or file -> pinned slot): wire_drop_vh is that
// copy without bytes 8..11 of every sample, so a sector crosses PCIe as 8 bytes per sample.  The bytes that are kept are
// not touched (no swap, no conversion: the GPU decodes them, bit-identically to Sector::fromByteArray).  The copy writes with
// NON-TEMPORAL stores: the slot is read next by the GPU's DMA engine and never again by the CPU.
//
// FillPool: the same copy (or a plain memcpy) of one sector split over T threads -- the caller is one of them; the helpers
// spin for a few microseconds between sectors before they sleep (a sector is ~100 us of copying: a condition-variable
// wake-up per helper and sector costs as much as the helper saves, tools/fillbench.cpp).
#ifndef WRP_HOST_WIRE_H
#define WRP_HOST_WIRE_H
#include <stddef.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#define WIRE_BYTES_PER_SAMPLE 12        // sector.cpp:52-62
#define WIRE8_BYTES_PER_SAMPLE 8        // ... without the VH pair

// dst8[8 i .. 8 i + 7] = src12[12 i .. 12 i + 7], i < samples.  The buffers must not overlap.
void wire_drop_vh(unsigned char *dst8, const unsigned char *src12, size_t samples);
void wire_drop_vh_portable(unsigned char *dst8, const unsigned char *src12, size_t samples);   // the reference loop (tests)
// memcpy into a pinned slot with non-temporal stores (the 12-byte feeder's copy): the GPU's DMA engine reads the slot next
// and the CPU never again, so the bytes go past the caches (see wire.cpp)
void wire_copy_to_pinned(char *dst, const char *src, size_t bytes);

class FillPool {
  public:
    explicit FillPool(int threads);
    ~FillPool();
    FillPool(const FillPool &) = delete;
    FillPool &operator=(const FillPool &) = delete;
    int threads() const { return n_; }
    // plain copy of `bytes` bytes
    void copy(char *dst, const char *src, size_t bytes);
    // `samples` 12-byte samples at src -> 8-byte samples at dst
    void drop_vh(char *dst8, const char *src12, size_t samples);

  private:
    void run_job();
    void part(int t);
    void worker(int t);
    const int n_;
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::atomic<long> gen_{0};      // job number: a helper works on job g when it sees gen_ == g
    std::atomic<int> left_{0};      // helpers that have not finished the current job
    std::atomic<int> asleep_{0};    // helpers waiting on cv_ (they spin first)
    bool stop_ = false;
    char *dst_ = nullptr;
    const char *src_ = nullptr;
    size_t units_ = 0;              // bytes (copy) or samples (drop_vh)
    bool drop_ = false;
};
#endif
