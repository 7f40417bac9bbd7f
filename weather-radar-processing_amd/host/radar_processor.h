// radar_processor.h -- the reference's RadarProcessor (radar_processor.h:14-96), finished, on top
// of the C ABI of include/wrp.h.  Same constructor, start() and set_comms(); the private methods
// keep the reference's names.  Differences, all deliberate:
//   * read_matrix does NOT decode on the CPU: the datagrams go straight into the engine's pinned
//     wire-format slot and are decoded on the GPU (wrp_submit_raw);
//   * perform_stage_1/2/3 are one asynchronous submit (the reference left 2 and 3 empty,
//     radar_processor.cu:239-247);
//   * copy_result_to_host + send_results wait for the slot's event instead of reading the result
//     buffer unsynchronised (gpu_1fp_streamcasc.cu:695-697);
//   * start() returns when the source ends or max_sectors is reached (the reference never returns);
//   * set_wire_bytes(8): the sector crosses PCIe WITHOUT its VH samples (WRP_FLAG_WIRE_8: no output reads them,
//     rpv2.cu:199-213): the copy that brings the bytes into the pinned slot drops them (wire.h).
#ifndef WRP_HOST_RADAR_PROCESSOR_H
#define WRP_HOST_RADAR_PROCESSOR_H
#include <stddef.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "udpbroadcast.h"
#include "wire.h"
#include "wrp.h"

#define NUM_BYTES_PER_SAMPLE (3 * 2 * 2)

// Several RadarProcessors -- one per GPU, each on its own host thread -- share ONE source (a socket
// delivers the sectors of a volume scan in acquisition order).  The turnstile hands the source to the
// processor that owns the next sector (sector s of every elevation belongs to processor s mod G,
// SURVEY 8e) and to nobody else; everything behind the read (H2D, kernels, D2H, egress) runs in
// parallel on the G GPUs.  No data-path exchange between the GPUs, hence no collective.
struct SectorTurnstile {
    std::mutex mu;
    std::condition_variable cv;
    long next = 0;        // global sequence number (elevation * n_sectors + sector) of the sector to be read next
    bool reading = false; // a processor is inside the (blocking) source call -- made WITHOUT the mutex held
    std::atomic<bool> ended{false};   // the source is exhausted (or failed): nobody reads any more (written under mu; rpv2.cpp's UDP retry loop reads it without)
    std::mutex sink_mu;   // frames of different GPUs leave through one sink, one frame at a time
};

class RadarProcessor {
  public:
    RadarProcessor(int num_sectors, int num_sweeps, int num_samples, int num_elevations, int num_streams);
    ~RadarProcessor();
    int start();
    void set_comms(int in_port, int *out_ports, int n_out);   // UDP: in_port raw sectors, out_ports[0] Zdb, [1] Zdr
    void set_unicast(const char *ipv4) { unicast_ = ipv4 ? ipv4 : ""; }   // products to this host instead of the broadcast address

    // hooks the reference does not have (tests, file replay, other transports)
    // one sector of wire bytes IN THE PROCESSOR'S FORMAT (wire_bytes() per sample) into buf; false = end.  A source that
    // delivers the 12-byte samples of sector.cpp:52-62 is wrapped with drop_vh_source() when the format is 8.
    typedef std::function<bool(char *buf, size_t bytes)> Source;
    typedef std::function<void(int which, int sector, int elevation, const unsigned char *frame, size_t bytes)> Sink;
    void set_source(Source s) { source_ = std::move(s); }
    void set_sink(Sink s) { sink_ = std::move(s); }
    void set_device(int d) { device_ = d; }
    void set_wire_bytes(int b) { wire_bytes_ = b == 8 ? 8 : 12; }   // before start() / set_comms()
    int wire_bytes() const { return wire_bytes_; }
    // a source of 12-byte samples as a source of 8-byte ones: the sector goes through a staging buffer of this processor,
    // the copy into `buf` drops VH
    Source drop_vh_source(Source src12);
    // this processor owns the sectors s with s % world == rank of every elevation (world GPUs, one
    // processor each); the processors of one scan share `turn`
    void set_shard(int rank, int world, SectorTurnstile *turn) { shard_rank_ = rank; shard_world_ = world; turn_ = turn; }
    void set_max_sectors(long n) { max_sectors_ = n; }
    void set_on_ready(std::function<void()> f) { on_ready_ = std::move(f); }   // called once the engine exists and the loop starts
    void set_frame_with_elevation(bool on) { with_elevation_ = on; }
    long sectors_done() const { return done_; }
    double processing_seconds() const { return seconds_; }   // wall clock of do_process (engine set-up excluded)
    // where the feeder thread spent that time: in the source (read_matrix), queuing work (wrp_submit_raw), waiting for a
    // slot (wrp_wait) and in the sink (send_results)
    struct Breakdown { double source = 0, submit = 0, wait = 0, sink = 0; };
    Breakdown breakdown() const { return spent_; }
    // sectors per second once the first kWarmSectors are through (a run's first sectors pay for the first use of every
    // kernel and of every pinned buffer: ~15 ms, a tenth of a 1000-sector run); 0 when the run was shorter
    static const long kWarmSectors = 128;
    double steady_rate() const { return steady_; }
    const char *last_error() const;

    const int input_ary_size, input_columns, input_rows, output_ary_size, output_columns, output_rows;

  private:
    const int n_sectors, n_sweeps, n_samples, n_elevations;
    static const int k_range_resolution = 30;
    static constexpr float k_calibration = 1941.05f;
    static const int ma_count = 7;
    const int n_streams;
    int current_sector = 0, current_elevation = 0, current_stream = 0;
    const int o_types = 2;

    wrp_handle eng_ = nullptr;
    int device_ = 0;
    int shard_rank_ = 0, shard_world_ = 1;
    SectorTurnstile *turn_ = nullptr;
    long laps_ = 0;       // completed passes over the elevations (the global sequence number keeps growing)
    long max_sectors_ = -1, done_ = 0;
    double seconds_ = 0;
    Breakdown spent_;
    double steady_ = 0;
    bool with_elevation_ = true;
    int status_ = 0;
    int wire_bytes_ = 12;
    std::vector<char> stage_;     // drop_vh_source / the UDP rows: 12-byte samples on their way into an 8-byte slot
    std::string unicast_;
    Source source_;
    std::function<void()> on_ready_;
    Sink sink_;
    std::unique_ptr<udpbroadcast::udpserver> server_;
    std::vector<std::unique_ptr<udpbroadcast::udpclient>> clients_;

    void generate_constants();
    void prepare_arys();
    void initialize_streams();
    void do_process();
    void destroy_streams();
    void destroy_arrays();

    bool read_matrix(int sector, int elevation, int stream);
    void copy_matrix_to_device(int sector, int elevation, int stream);
    void perform_stage_1(int stream);
    void perform_stage_2(int stream);
    void perform_stage_3(int stream);
    void advance();
    void copy_result_to_host(int sector, int elevation, int stream);
    void send_results(int sector, int elevation);
};
#endif
