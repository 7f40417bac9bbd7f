// host_capi.cpp -- C-linkage view of the host types for tests and non-C++ callers.
#include <string.h>

#include <chrono>
#include <memory>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "dimension.h"
#include "floats.h"
#include "framing.h"
#include "sector.h"
#include "tcp.h"
#include "wire.h"

extern "C" {

void wrph_sector_from_bytes(char *buff, int sweeps, int samples, short *hh, short *vv, short *vh)
{
    Sector s(sweeps, samples);
    s.fromByteArray(buff);
    const size_t cnt = sizeof(short) * 2 * (size_t)sweeps * samples;
    memcpy(hh, s.hh, cnt);
    memcpy(vv, s.vv, cnt);
    memcpy(vh, s.vh, cnt);
}
// Sector::read from a stream holding `nbytes` bytes (sector.cpp:26-45 in the reference).  The sector is zero-filled
// first, so a short stream shows as zeros behind the last whole sample.
void wrph_sector_read(const char *bytes, size_t nbytes, int sweeps, int samples, short *hh, short *vv, short *vh)
{
    Sector s(sweeps, samples);
    const size_t cnt = sizeof(short) * 2 * (size_t)sweeps * samples;
    memset(s.hh, 0, cnt);
    memset(s.vv, 0, cnt);
    memset(s.vh, 0, cnt);
    std::istringstream in(std::string(bytes, nbytes));
    s.read(in);
    memcpy(hh, s.hh, cnt);
    memcpy(vv, s.vv, cnt);
    memcpy(vh, s.vh, cnt);
}
// WRP_FLAG_WIRE_8's feeder side: 12-byte samples -> 8-byte samples (VH dropped); threads > 1: through a FillPool; portable != 0: the plain loop
void wrph_wire_drop_vh(unsigned char *dst8, const unsigned char *src12, size_t samples, int threads, int portable)
{
    if (portable) { wire_drop_vh_portable(dst8, src12, samples); return; }
    if (threads <= 1) { wire_drop_vh(dst8, src12, samples); return; }
    FillPool pool(threads);
    for (int rep = 0; rep < 3; rep++) pool.drop_vh((char *)dst8, (const char *)src12, samples);     // (the pool is reused per sector)
}
// FillPool under load: `jobs` sectors of `samples` samples through ONE pool of `threads` threads, alternating the VH-dropping
// copy and the plain copy, a pause of `pause_us` every 64 jobs (so that the helpers fall asleep and are woken again);
// returns the number of jobs whose output differed from the single-threaded copy
long wrph_fill_pool_stress(int threads, long jobs, size_t samples, int pause_us)
{
    std::vector<unsigned char> src(samples * 12 + 64), want8(samples * 8), got(samples * 12 + 64);
    FillPool pool(threads);
    long bad = 0;
    for (long j = 0; j < jobs; j++) {
        for (size_t k = 0; k < src.size(); k += 97) src[k] = (unsigned char)(j * 31 + k);     // a few bytes change per job
        if (j & 1) {
            pool.copy((char *)got.data(), (const char *)src.data(), samples * 12);
            bad += memcmp(got.data(), src.data(), samples * 12) != 0;
        } else {
            wire_drop_vh_portable(want8.data(), src.data(), samples);
            pool.drop_vh((char *)got.data(), (const char *)src.data(), samples);
            bad += memcmp(got.data(), want8.data(), samples * 8) != 0;
        }
        if (pause_us > 0 && (j & 63) == 63) std::this_thread::sleep_for(std::chrono::microseconds(pause_us));
    }
    return bad;
}
void wrph_aftoab(float *af, size_t n, unsigned char *ab) { aftoab(af, n, ab); }
void wrph_abtoaf(unsigned char *ab, size_t n, float *af) { abtoaf(ab, n, af); }
int wrph_dim3_at_depth(int w, int h, int d, int x, int y, int depth) { return Dimension3(w, h, d).at_depth(x, y, depth); }
int wrph_dim4_copy_at_depth(int w, int h, int c, int d, int x, int y, int copy, int depth)
{
    return Dimension4(w, h, c, d).copy_at_depth(x, y, copy, depth);
}
size_t wrph_frame_result(const float *zdb_zdr, int gates, int sector, int elevation, int which, int with_elevation,
                         unsigned char *out)
{
    return frame_result(zdb_zdr, gates, sector, elevation, which, with_elevation != 0, out);
}

// tcp.h loop-back exercise: a tcpserver thread on `port` receives `messages` messages of `length` bytes
// from a tcpclient in the calling thread.  Returns 0 when every message arrived intact and every
// sendit() saw its acknowledgement; a negative code says which step failed.
int wrph_tcp_loopback(int port, int messages, int length)
{
    std::vector<std::vector<char>> got(messages, std::vector<char>(length));
    int server_rc = 0;
    std::thread server([&] {
        try {
            tcp::tcpserver s(port);
            for (int k = 0; k < messages; k++)
                if (s.recv(got[k].data(), (size_t)length) != length) { server_rc = -2; return; }
        } catch (const char *) { server_rc = -1; }
    });
    int rc = 0;
    try {
        // the server may not be listening yet: connect refused -> retry for up to two seconds
        std::unique_ptr<tcp::tcpclient> c;
        for (int attempt = 0; attempt < 200 && !c; attempt++) {
            try { c.reset(new tcp::tcpclient(port)); }
            catch (const char *) { std::this_thread::sleep_for(std::chrono::milliseconds(10)); }
        }
        if (!c) rc = -3;
        std::vector<char> msg(length);
        for (int k = 0; k < messages && rc == 0; k++) {
            for (int i = 0; i < length; i++) msg[i] = (char)(31 * i + 7 * k + 1);
            if (c->sendit(msg.data(), (size_t)length) != 0) rc = -4;
        }
    } catch (const char *) { rc = -5; }
    if (rc == -3) {   // never connected: the server thread either failed to bind (and has ended) or waits in accept
        server.detach();
        return server_rc != 0 ? server_rc : rc;
    }
    server.join();
    if (rc == 0 && server_rc != 0) rc = server_rc;
    for (int k = 0; k < messages && rc == 0; k++)
        for (int i = 0; i < length; i++)
            if (got[k][i] != (char)(31 * i + 7 * k + 1)) { rc = -6; break; }
    return rc;
}

} // extern "C"
