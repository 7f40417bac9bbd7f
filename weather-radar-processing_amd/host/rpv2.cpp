// rpv2.cpp -- entry point with the reference's name (Makefile:4 builds `rpv2`; main.cpp:3-16 is
// the class-based variant).  usage:
//   rpv2 [num_streams] [--in udp:PORT | file:PATH | synthetic | synthetic:copy[:T]] [--bind-numa] [--out udp:PORT_ZDB,PORT_ZDR[@IPV4] | file:PATH | none]
//        [--sectors N] [--device D | --devices D0,D1,...] [--no-elevation] [--scan SECTORS,ELEVATIONS] [--wire8]
// Defaults reproduce main.cpp:10-15: 143 sectors x 9 elevations of 1024 x 512, UDP 19001 in,
// 19002 / 19003 out (broadcast, as the reference; @IPV4 sends the products to one host instead).  file: input is a concatenation of wire-format sectors (12 bytes/sample),
// file: output a concatenation of frames, Zdb then Zdr per sector.  Frame header: [sector BE16] for udp: products
// (read_single.cc:510-517), [sector BE16][elevation BE16] (rpv2.cu:631-661) for file: -- keyed on the OUTPUT endpoint;
// --no-elevation forces the short header.
//
// --devices: ONE host thread and ONE engine handle per listed GPU (a device may be listed twice to
// rehearse on a smaller box); sector s of every elevation goes to GPU number s mod G of the list; the
// threads take turns on the one source in acquisition order (SectorTurnstile) and run everything behind
// the read -- pinned H2D on the GPU's slot cascade, decode, kernels, D2H, egress -- in parallel.  No
// collective: sectors are independent.  With more than one GPU the frames of different
// GPUs interleave on the output (each frame names its sector and elevation).
// --in synthetic: the pinned slots keep whatever they hold (zeros at start): transport + GPU pipeline
// rate without a real source.  The run's wall-clock rate is printed on stderr.
// --in synthetic:copy[:T]: every sector is COPIED into its pinned slot from an ordinary pageable buffer of the
// feeder thread (6 MiB per sector, as a socket delivers it), by the feeder thread and T - 1 helpers (default
// T = 1): the end-to-end rate WITH the host's share of the work.  Each GPU thread has a source of its own
// (no turnstile), and --sectors counts per GPU.
// --bind-numa: every GPU thread (and its helpers) runs on the CPUs of the NUMA node its GPU hangs on.
// --wire8: the sectors cross PCIe without their VH samples (WRP_FLAG_WIRE_8: 8 bytes per sample instead of 12; no output
// reads VH, rpv2.cu:199-213): the copy that brings a sector into its pinned slot -- from the socket's row buffer, the
// file's staging buffer or synthetic:copy's pageable buffer -- drops bytes 8..11 of every sample (host/wire.h).  The
// input (socket, file) is the reference's 12-byte format either way.
// --fill-bench: no sector is processed; times the host's copy of a sector from a pageable buffer into the engine's pinned
// slots by itself -- plain copy and VH-dropping copy, 1 / 2 / 4 / 8 threads, with and without --bind-numa -- and prints
// microseconds per sector and GB/s of bytes read (what bounds bench.py's end_to_end.with_host_fill).
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include <sched.h>

#include <atomic>
#include <condition_variable>
#include <mutex>

#include "radar_processor.h"
#include "wire.h"

namespace {

// the calling thread (and every thread it starts later) onto the CPUs of the GPU's NUMA node; false = left alone
bool bind_to_gpu_numa(int device)
{
    const int node = wrp_device_numa_node(device);
    if (node < 0) return false;
    char path[96], list[4096] = {0};
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = fopen(path, "r");
    if (!f) return false;
    const bool got = fgets(list, sizeof list, f) != nullptr;
    fclose(f);
    if (!got) return false;
    cpu_set_t want, have, set;
    CPU_ZERO(&want);
    for (char *tok = strtok(list, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
        int a = 0, b = 0;
        const int k = sscanf(tok, "%d-%d", &a, &b);
        if (k == 1) b = a;
        if (k >= 1) for (int c = a; c <= b && c < CPU_SETSIZE; c++) CPU_SET(c, &want);
    }
    if (sched_getaffinity(0, sizeof have, &have) != 0) return false;
    CPU_AND(&set, &want, &have);
    if (CPU_COUNT(&set) == 0) return false;       // the node's CPUs are outside what this process may use
    return sched_setaffinity(0, sizeof set, &set) == 0;
}

// --fill-bench (see the header): the copy into the pinned slots by itself
int run_fill_bench(int device, int slots, bool bind_numa)
{
    const bool bound = bind_numa && bind_to_gpu_numa(device);
    const size_t samples = (size_t)1024 * 512;
    std::vector<char> pageable(samples * WIRE_BYTES_PER_SAMPLE, 3);
    for (int wb : {12, 8}) {
        wrp_config cfg;
        wrp_default_config(&cfg);
        cfg.n_slots = slots;
        cfg.n_sectors = 1;
        cfg.n_elevations = 1;
        if (wb == 8) cfg.flags |= WRP_FLAG_WIRE_8;
        wrp_handle h = nullptr;
        if (wrp_create(&cfg, device, &h) != WRP_OK) { fprintf(stderr, "fill-bench: wrp_create failed\n"); return 1; }
        std::vector<char *> slot(slots);
        for (int s = 0; s < slots; s++) {
            void *p = nullptr;
            size_t bytes = 0;
            if (wrp_pinned_raw_slot(h, s, &p, &bytes) != WRP_OK || bytes != samples * wb) { fprintf(stderr, "fill-bench: slot\n"); return 1; }
            slot[s] = (char *)p;
        }
        for (int T : {1, 2, 4, 8}) {
            FillPool pool(T);
            const int reps = 400;
            double best = 1e30;
            for (int pass = 0; pass < 3; pass++) {
                const auto t0 = std::chrono::steady_clock::now();
                for (int k = 0; k < reps; k++) {
                    if (wb == 8) pool.drop_vh(slot[k % slots], pageable.data(), samples);
                    else pool.copy(slot[k % slots], pageable.data(), samples * WIRE_BYTES_PER_SAMPLE);
                }
                const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
                if (dt < best) best = dt;
            }
            fprintf(stderr, "fill-bench: %2d B/sample into the pinned slot, %d thread(s)%s: %7.1f us per sector = %6.0f sectors/s, %5.1f GB/s read + %5.1f GB/s written\n",
                    wb, T, bound ? ", NUMA-bound" : "", best * 1e6, 1.0 / best, samples * 12 / best / 1e9, samples * wb / best / 1e9);
        }
        wrp_destroy(h);
    }
    return 0;
}

} // namespace

int main(int argc, char **argv)
{
    int num_streams = 2;
    long sectors = -1;
    bool with_elev = true, with_elev_set = false, bind_numa = false, wire8 = false, fill_bench = false;
    int scan_sectors = 143, scan_elevations = 9;
    std::vector<int> devices{0};
    std::string in = "udp:19001", out = "udp:19002,19003";
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a == "--in" && i + 1 < argc) in = argv[++i];
        else if (a == "--out" && i + 1 < argc) out = argv[++i];
        else if (a == "--sectors" && i + 1 < argc) sectors = atol(argv[++i]);
        else if (a == "--device" && i + 1 < argc) devices = {atoi(argv[++i])};
        else if (a == "--devices" && i + 1 < argc) {
            devices.clear();
            for (char *tok = strtok(argv[++i], ","); tok; tok = strtok(nullptr, ",")) devices.push_back(atoi(tok));
            if (devices.empty()) devices = {0};
        } else if (a == "--scan" && i + 1 < argc) {
            if (sscanf(argv[++i], "%d,%d", &scan_sectors, &scan_elevations) != 2 || scan_sectors < 1 || scan_elevations < 1) {
                fprintf(stderr, "--scan SECTORS,ELEVATIONS\n");
                return 2;
            }
        } else if (a == "--no-elevation") { with_elev = false; with_elev_set = true; }
        else if (a == "--bind-numa") bind_numa = true;
        else if (a == "--wire8") wire8 = true;
        else if (a == "--fill-bench") fill_bench = true;
        else if (a[0] != '-') { num_streams = atoi(a.c_str()); if (num_streams < 1) num_streams = 1; }
        else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    if (fill_bench) return run_fill_bench(devices[0], num_streams < 4 ? 4 : num_streams, bind_numa);
    const int G = (int)devices.size();
    std::vector<std::unique_ptr<RadarProcessor>> procs;
    SectorTurnstile turn;
    for (int g = 0; g < G; g++) {
        procs.emplace_back(new RadarProcessor(scan_sectors, 1024, 512, scan_elevations, num_streams));
        procs[g]->set_device(devices[g]);
        procs[g]->set_wire_bytes(wire8 ? 8 : 12);
        procs[g]->set_max_sectors(sectors);
        if (G > 1) procs[g]->set_shard(g, G, in.rfind("synthetic", 0) == 0 ? nullptr : &turn);   // synthetic: a source per GPU
    }
    const bool copy_source = in.rfind("synthetic:copy", 0) == 0;
    const int fill_threads = copy_source && in.size() > 15 ? atoi(in.c_str() + 15) : 1;
    std::vector<std::unique_ptr<FillPool>> pools(G);
    std::vector<std::vector<char>> pageable(G);
    FILE *fin = nullptr, *fout = nullptr;
    std::unique_ptr<udpbroadcast::udpserver> server;
    std::vector<std::unique_ptr<udpbroadcast::udpclient>> clients;
    try {
        if (in.rfind("udp:", 0) == 0 && out.rfind("udp:", 0) == 0 && G == 1) {
            int ports[2] = {19002, 19003};
            sscanf(out.c_str() + 4, "%d,%d", &ports[0], &ports[1]);
            if (out.find('@') != std::string::npos) procs[0]->set_unicast(out.c_str() + out.find('@') + 1);
            procs[0]->set_comms(atoi(in.c_str() + 4), ports, 2);
        } else {
            RadarProcessor::Source src;
            RadarProcessor::Sink sink;
            if (in.rfind("file:", 0) == 0) {
                fin = fopen(in.c_str() + 5, "rb");
                if (!fin) { perror("input"); return 2; }
                src = [fin](char *buf, size_t bytes) { return fread(buf, 1, bytes, fin) == bytes; };
            } else if (in == "synthetic" || copy_source) {
                src = [](char *, size_t) { return true; };     // synthetic:copy: per GPU thread, set where the thread starts
            } else if (in.rfind("udp:", 0) == 0) {
                server.reset(new udpbroadcast::udpserver(atoi(in.c_str() + 4)));
                udpbroadcast::udpserver *sv = server.get();
                // several GPU threads share the socket: a reader waiting on a source that has gone quiet looks at the
                // turnstile every 200 ms, so that the failure of another GPU's thread ends the run instead of hanging it
                if (G > 1) sv->set_timeout_ms(200);
                SectorTurnstile *tn = G > 1 ? &turn : nullptr;
                src = [sv, tn](char *buf, size_t bytes) {   // one datagram per range row (read_single.cc:145-148)
                    const size_t row = (size_t)NUM_BYTES_PER_SAMPLE * 512;
                    for (size_t off = 0; off < bytes;) {
                        const int got = sv->recv(buf + off, row);
                        if (got == (int)row) { off += row; continue; }
                        if (got < 0 && (errno == EAGAIN || errno == EWOULDBLOCK) && tn && !tn->ended) continue;
                        return false;
                    }
                    return true;
                };
            } else { fprintf(stderr, "unknown --in %s\n", in.c_str()); return 2; }
            if (out.rfind("file:", 0) == 0) {
                fout = fopen(out.c_str() + 5, "wb");
                if (!fout) { perror("output"); return 2; }
                sink = [fout](int, int, int, const unsigned char *f, size_t n) { fwrite(f, 1, n, fout); };
            } else if (out.rfind("udp:", 0) == 0) {
                if (!with_elev_set) with_elev = false, with_elev_set = true;   // UDP products carry the 2-byte header (read_single.cc:510-517)
                int ports[2] = {19002, 19003};
                sscanf(out.c_str() + 4, "%d,%d", &ports[0], &ports[1]);
                const size_t at = out.find('@');
                for (int k = 0; k < 2; k++)
                    clients.emplace_back(at == std::string::npos ? new udpbroadcast::udpclient(ports[k])
                                                                 : new udpbroadcast::udpclient(ports[k], out.c_str() + at + 1));
                auto *cl = &clients;
                sink = [cl](int which, int, int, const unsigned char *f, size_t n) { (*cl)[which]->send((const char *)f, n); };
            } else if (out != "none") { fprintf(stderr, "unknown --out %s\n", out.c_str()); return 2; }
            for (auto &p : procs) {
                // file: and udp: deliver the reference's 12-byte samples; synthetic sources fill whatever the slot holds
                p->set_source(wire8 && in.rfind("synthetic", 0) != 0 ? p->drop_vh_source(src) : src);
                if (sink) p->set_sink(sink);
            }
        }
    } catch (const char *msg) {
        fprintf(stderr, "socket: %s\n", msg);
        return 3;
    }
    if (with_elev_set) for (auto &p : procs) p->set_frame_with_elevation(with_elev);
    procs[0]->set_on_ready([] { fprintf(stderr, "rpv2: ready\n"); fflush(stderr); });

    std::vector<int> rcs(G, 0);
    std::vector<int> bound(G, 0);
    // runs on the GPU's own thread: NUMA binding first, so that the helpers and the pageable buffer (first touch) follow it
    auto prepare = [&](int g) {
        if (bind_numa) bound[g] = bind_to_gpu_numa(devices[g]);
        if (!copy_source) return;
        const size_t bytes = (size_t)NUM_BYTES_PER_SAMPLE * 1024 * 512;
        pageable[g].assign(bytes, (char)(g + 1));
        pools[g].reset(new FillPool(fill_threads));
        FillPool *pool = pools[g].get();
        const char *from = pageable[g].data();
        if (wire8)   // n = 8 bytes per sample of the slot; the pageable buffer holds the 12-byte samples
            procs[g]->set_source([pool, from](char *buf, size_t n) { pool->drop_vh(buf, from, n / WIRE8_BYTES_PER_SAMPLE); return true; });
        else
            procs[g]->set_source([pool, from, bytes](char *buf, size_t n) { pool->copy(buf, from, n < bytes ? n : bytes); return true; });
    };
    const auto t0 = std::chrono::steady_clock::now();
    if (G == 1) {
        prepare(0);
        rcs[0] = procs[0]->start();
    } else {
        std::vector<std::thread> threads;
        for (int g = 0; g < G; g++)
            threads.emplace_back([&, g] {
                prepare(g);
                try { rcs[g] = procs[g]->start(); } catch (const char *msg) { fprintf(stderr, "socket: %s\n", msg); rcs[g] = 3; }
                if (rcs[g]) {   // a GPU that failed must not leave the others waiting for its turn
                    std::lock_guard<std::mutex> lk(turn.mu);
                    turn.ended = true;
                    turn.cv.notify_all();
                }
            });
        for (auto &t : threads) t.join();
    }
    (void)t0;
    double dt = 1e-9;   // the slowest GPU's processing time (engine set-up excluded)
    for (auto &p : procs) dt = p->processing_seconds() > dt ? p->processing_seconds() : dt;
    if (fin) fclose(fin);
    if (fout) fclose(fout);
    long total = 0;
    int rc = 0;
    for (int g = 0; g < G; g++) {
        total += procs[g]->sectors_done();
        if (rcs[g]) { fprintf(stderr, "rpv2: GPU %d: %s\n", devices[g], procs[g]->last_error()); rc = 1; }
        else if (G > 1) fprintf(stderr, "rpv2: GPU %d (shard %d of %d): %ld sectors\n", devices[g], g, G, procs[g]->sectors_done());
    }
    pools.clear();
    if (!rc) fprintf(stderr, "rpv2: %ld sectors processed in %.3f s (%.0f sectors/s end to end, %d GPU thread%s, %d slots each%s%s)\n", total,
                     dt, total / dt, G, G > 1 ? "s" : "", num_streams,
                     copy_source ? (std::string(", host fill by ") + std::to_string(fill_threads) + " thread(s) per GPU" + (wire8 ? ", VH dropped (8 bytes per sample)" : "")).c_str() : "",
                     bind_numa ? (bound[0] ? ", NUMA-bound" : ", NUMA binding not possible") : "");
    if (!rc && total > 0) {      // the feeder thread's time per sector, by what it was doing (GPU thread 0)
        const RadarProcessor::Breakdown b = procs[0]->breakdown();
        const double k = 1e6 / (double)procs[0]->sectors_done();
        fprintf(stderr, "rpv2: feeder thread, us per sector: source %.1f, submit %.1f, wait %.1f, sink %.1f (of %.1f)\n", b.source * k, b.submit * k,
                b.wait * k, b.sink * k, procs[0]->processing_seconds() * k);
        double steady = 0;     // all GPU threads, each past its first sectors (the first use of kernels and pinned buffers excluded)
        for (auto &p : procs) steady += p->steady_rate();
        if (steady > 0) fprintf(stderr, "rpv2: steady state (each GPU thread past its first %ld sectors): %.0f sectors/s\n", RadarProcessor::kWarmSectors, steady);
    }
    return rc;
}
