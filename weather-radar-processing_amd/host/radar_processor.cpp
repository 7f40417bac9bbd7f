// radar_processor.cpp -- see radar_processor.h.  Control flow follows rpv2.cu:665-683 (do_process)
// and radar_processor.cu:48-57 (start).
#include "radar_processor.h"

#include <stdio.h>
#include <string.h>

#include <chrono>

#include "framing.h"

RadarProcessor::RadarProcessor(int num_sectors, int num_sweeps, int num_samples, int num_elevations, int num_streams)
    : input_ary_size(num_samples * num_sweeps), input_columns(num_samples), input_rows(num_sweeps),
      output_ary_size(2 * (num_sweeps / 2)), output_columns(2), output_rows(num_sweeps / 2),
      n_sectors(num_sectors), n_sweeps(num_sweeps), n_samples(num_samples), n_elevations(num_elevations),
      n_streams(num_streams < 1 ? 1 : num_streams)
{
}

RadarProcessor::~RadarProcessor()
{
    if (eng_) wrp_destroy(eng_);
}

RadarProcessor::Source RadarProcessor::drop_vh_source(Source src12)
{
    return [this, src12](char *buf, size_t bytes) {
        const size_t samples = bytes / WIRE8_BYTES_PER_SAMPLE;
        stage_.resize(samples * WIRE_BYTES_PER_SAMPLE);
        if (!src12(stage_.data(), stage_.size())) return false;
        wire_drop_vh((unsigned char *)buf, (const unsigned char *)stage_.data(), samples);
        return true;
    };
}

const char *RadarProcessor::last_error() const
{
    return status_ ? wrp_strerror(status_) : "";
}

void RadarProcessor::set_comms(int in_port, int *out_ports, int n_out)
{
    // radar_processor.cu:59-68: one UDP server for ingest, one client per product
    server_.reset(new udpbroadcast::udpserver(in_port));
    clients_.clear();
    for (int i = 0; i < n_out; i++)
        clients_.emplace_back(unicast_.empty() ? new udpbroadcast::udpclient(out_ports[i])
                                               : new udpbroadcast::udpclient(out_ports[i], unicast_.c_str()));
    source_ = [this](char *buf, size_t bytes) {
        // one datagram per range row: m datagrams of 12*n bytes (read_single.cc:145-148)
        const size_t row = (size_t)NUM_BYTES_PER_SAMPLE * n_samples;
        if (wire_bytes_ == 8) {     // the row lands in a 6 KiB buffer and goes into the slot without its VH samples
            stage_.resize(row);
            const size_t row8 = (size_t)WIRE8_BYTES_PER_SAMPLE * n_samples;
            for (size_t off = 0; off < bytes; off += row8) {
                if (server_->recv(stage_.data(), row) != (int)row) return false;
                wire_drop_vh((unsigned char *)buf + off, (const unsigned char *)stage_.data(), n_samples);
            }
            return true;
        }
        for (size_t off = 0; off < bytes; off += row)
            if (server_->recv(buf + off, row) != (int)row) return false;
        return true;
    };
    with_elevation_ = false;   // the UDP products carry the 2-byte header (read_single.cc:510-517)
    sink_ = [this](int which, int, int, const unsigned char *frame, size_t bytes) {
        if (which < (int)clients_.size()) clients_[which]->send((const char *)frame, bytes);
    };
}

int RadarProcessor::start()
{
    if (!source_) return status_ = WRP_ERR_STATE;
    generate_constants();
    prepare_arys();
    initialize_streams();
    if (status_ == WRP_OK && on_ready_) on_ready_();
    const auto t0 = std::chrono::steady_clock::now();
    if (status_ == WRP_OK) do_process();
    seconds_ = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    destroy_streams();
    destroy_arrays();
    return status_;
}

// generate_constants + prepare_arys + initialize_streams are ONE call in the C ABI (wrp_create
// builds the window / twiddle / MA tables, the per-slot buffers and the streams); the three
// reference stages are kept as names so the call graph reads like rpv2.cu:738-742.
void RadarProcessor::generate_constants() {}
void RadarProcessor::prepare_arys() {}

void RadarProcessor::initialize_streams()
{
    wrp_config cfg;
    wrp_default_config(&cfg);
    cfg.m = n_sweeps;
    cfg.n = n_samples;
    cfg.channels = 2;               // VH is on the wire but feeds no product (rpv2.cu:199-213)
    cfg.n_slots = n_streams;
    cfg.n_sectors = n_sectors;
    cfg.n_elevations = n_elevations;
    cfg.ma_count = ma_count;
    cfg.k_range_resolution = (float)k_range_resolution;
    cfg.k_calibration = k_calibration;
    if (wire_bytes_ == 8) cfg.flags |= WRP_FLAG_WIRE_8;
    status_ = wrp_create(&cfg, device_, &eng_);
}

bool RadarProcessor::read_matrix(int sector, int elevation, int stream)
{
    void *raw = nullptr;
    size_t bytes = 0;
    status_ = wrp_pinned_raw_slot(eng_, stream, &raw, &bytes);
    if (status_ != WRP_OK) return false;
    if (!turn_) return source_((char *)raw, bytes);      // no CPU decode, no int16->float scatter (rpv2.cu:364-383)
    // sharded scan: wait until it is this sector's turn on the shared source, read it straight into
    // this GPU's pinned slot, pass the turn on
    const long seq = (laps_ * n_elevations + elevation) * (long)n_sectors + sector;
    std::unique_lock<std::mutex> lk(turn_->mu);
    turn_->cv.wait(lk, [&] { return turn_->ended || (turn_->next == seq && !turn_->reading); });
    if (turn_->ended) return false;
    if (max_sectors_ >= 0 && seq >= max_sectors_) {       // the scan's sector budget (a GLOBAL count) is used up
        turn_->ended = true;
        turn_->cv.notify_all();
        return false;
    }
    // the read itself may block for a whole sector (a UDP recv): not under the mutex, or a thread that has to
    // announce its failure (rpv2.cpp) could not take it
    turn_->reading = true;
    lk.unlock();
    const bool ok = source_((char *)raw, bytes);
    lk.lock();
    turn_->reading = false;
    if (ok) turn_->next = seq + 1; else turn_->ended = true;
    turn_->cv.notify_all();
    return ok;
}

void RadarProcessor::copy_matrix_to_device(int sector, int elevation, int stream)
{
    // H2D + decode + stages 1..3 + D2H are queued together on the slot's stream
    status_ = wrp_submit_raw(eng_, stream, sector, elevation);
}
void RadarProcessor::perform_stage_1(int) {}
void RadarProcessor::perform_stage_2(int) {}
void RadarProcessor::perform_stage_3(int) {}

void RadarProcessor::advance()
{
    // rpv2.cu:572-579; sharded: this processor's next sector is `world` further on
    current_sector += shard_world_;
    if (current_sector >= n_sectors) {
        current_sector = shard_rank_;
        current_elevation = (current_elevation + 1) % n_elevations;
        if (current_elevation == 0) laps_++;
    }
    current_stream = (current_stream + 1) % n_streams;
}

void RadarProcessor::copy_result_to_host(int, int, int stream)
{
    status_ = wrp_wait(eng_, stream);
}

void RadarProcessor::send_results(int sector, int elevation)
{
    // rpv2.cu:620-663 copies each product out of the result table, swaps every float to big-endian on the CPU (aftoab) and
    // prepends the header.  Here the GPU has written both products wire-ready into the pinned frame table:
    // the frame is handed to the sink where it lies.
    if (!sink_) return;
    std::unique_lock<std::mutex> lk;
    if (turn_) lk = std::unique_lock<std::mutex>(turn_->sink_mu);
    for (int which = 0; which < 2; which++) {
        const unsigned char *frame = nullptr;
        size_t n = 0;
        status_ = wrp_result_frame(eng_, sector, elevation, which, with_elevation_ ? 1 : 0, &frame, &n);
        if (status_ != WRP_OK) return;
        sink_(which, sector, elevation, frame, n);
    }
}

void RadarProcessor::do_process()
{
    // rpv2.cu:665-683 with up to n_streams sectors in flight: while the GPU works on slot s the
    // host already receives the next sector into slot s+1
    struct InFlight { int sector, elevation, stream; };
    std::vector<InFlight> q;
    bool more = true;
    current_sector = shard_rank_;
    if (current_sector >= n_sectors) more = false;          // more GPUs than sectors: nothing to do here
    const bool own_budget = turn_ == nullptr;               // sharded: the budget is global, read_matrix enforces it
    typedef std::chrono::steady_clock clk;
    auto since = [](clk::time_point t0) { return std::chrono::duration<double>(clk::now() - t0).count(); };
    clk::time_point t_warm;
    while (status_ == WRP_OK && (more || !q.empty())) {
        if (more && (int)q.size() < n_streams && (!own_budget || max_sectors_ < 0 || done_ + (long)q.size() < max_sectors_)) {
            auto t0 = clk::now();
            const bool got = read_matrix(current_sector, current_elevation, current_stream);
            spent_.source += since(t0);
            if (got && status_ == WRP_OK) {
                t0 = clk::now();
                copy_matrix_to_device(current_sector, current_elevation, current_stream);
                spent_.submit += since(t0);
                perform_stage_1(current_stream);
                perform_stage_2(current_stream);
                perform_stage_3(current_stream);
                q.push_back({current_sector, current_elevation, current_stream});
                advance();
                continue;
            }
            more = false;
            continue;
        }
        if (q.empty()) break;
        const InFlight f = q.front();
        q.erase(q.begin());
        auto t0 = clk::now();
        copy_result_to_host(f.sector, f.elevation, f.stream);
        spent_.wait += since(t0);
        if (status_ != WRP_OK) break;
        t0 = clk::now();
        send_results(f.sector, f.elevation);
        spent_.sink += since(t0);
        done_++;
        if (done_ == kWarmSectors) t_warm = clk::now();
        if (own_budget && max_sectors_ >= 0 && done_ + (long)q.size() >= max_sectors_) more = false;
    }
    if (done_ > kWarmSectors) steady_ = (double)(done_ - kWarmSectors) / since(t_warm);
}

void RadarProcessor::destroy_streams() {}

void RadarProcessor::destroy_arrays()
{
    if (eng_) wrp_destroy(eng_);
    eng_ = nullptr;
}
