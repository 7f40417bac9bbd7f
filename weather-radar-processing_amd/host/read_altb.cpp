// read_altb.cpp -- the reference's CPU program `read` (read.cc) with the GPU behind it: same input, same output.
//
//   read_altb [--m M] [--n N] [--device D] < in/00iq.altb > out/99result.out
//
// Input (read.cc:106-123): whitespace-separated text, m*n pairs "I Q" of HH in row-major order (i*n + j), then m*n
// pairs of VV.  A file may hold several sectors one after the other; the multi-sector variant of the format
// (gpu_1fp_streamreordered.cu:290-302, 418-430) puts ONE extra token -- the sector id, read and discarded -- in front
// of every sector: accepted when the token count says so.  Output: one line "zdb zdr" per range gate < m/2 and sector,
// as the line read.cc:344 prints when it is not commented out (ostream default formatting: out/99result.cpu.out).
// Sectors go through the engine's pinned planar slots two at a time (wrp_submit: H2D + chain + D2H, asynchronous).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <iostream>
#include <string>
#include <vector>

#include "wrp.h"

int main(int argc, char **argv)
{
    int m = 1024, n = 512, device = 0;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a == "--m" && i + 1 < argc) m = atoi(argv[++i]);
        else if (a == "--n" && i + 1 < argc) n = atoi(argv[++i]);
        else if (a == "--device" && i + 1 < argc) device = atoi(argv[++i]);
        else { fprintf(stderr, "usage: read_altb [--m M] [--n N] [--device D] < sectors.altb\n"); return 2; }
    }
    std::ios::sync_with_stdio(false);
    std::vector<double> tok;
    tok.reserve((size_t)4 * m * n + 1);
    for (double x; std::cin >> x;) tok.push_back(x);
    const size_t per = (size_t)4 * m * n;   // I, Q of HH and of VV
    size_t stride = 0;
    if (!tok.empty() && tok.size() % per == 0) stride = per;                 // read.cc's format
    else if (!tok.empty() && tok.size() % (per + 1) == 0) stride = per + 1;   // a sector id in front of every sector
    if (!stride) {
        fprintf(stderr, "read_altb: %zu numbers on stdin; expected a multiple of %zu (or of %zu with sector ids)\n", tok.size(), per,
                per + 1);
        return 2;
    }
    const long sectors = (long)(tok.size() / stride);

    wrp_config cfg;
    wrp_default_config(&cfg);
    cfg.m = m;
    cfg.n = n;
    cfg.channels = 2;
    cfg.n_slots = 2;
    cfg.n_sectors = 2;      // the result table is reused: two sectors in flight
    cfg.n_elevations = 1;
    wrp_handle eng = nullptr;
    int rc = wrp_create(&cfg, device, &eng);
    if (rc) { fprintf(stderr, "read_altb: %s (%s)\n", wrp_strerror(rc), eng ? wrp_last_hip_error(eng) : ""); return 1; }

    auto emit = [&](long s) -> int {
        const int slot = (int)(s & 1);
        int r = wrp_wait(eng, slot);
        if (r) return r;
        const float *z = nullptr;
        r = wrp_result(eng, slot, 0, &z);
        if (r) return r;
        for (int i = 0; i < m / 2; i++) std::cout << z[2 * i] << " " << z[2 * i + 1] << "\n";
        return 0;
    };
    for (long s = 0; s < sectors && !rc; s++) {
        const int slot = (int)(s & 1);
        if (s >= 2) rc = emit(s - 2);            // the slot's previous sector leaves before the slot is refilled
        if (rc) break;
        float *p = nullptr;
        size_t bytes = 0;
        rc = wrp_pinned_slot(eng, slot, (void **)&p, &bytes);
        if (rc) break;
        const double *src = tok.data() + (size_t)s * stride + (stride - per);
        for (size_t k = 0; k < per; k++) p[k] = (float)src[k];   // [channel][i][j][re, im]: exactly the order of the text
        rc = wrp_submit(eng, slot, slot, 0);
    }
    for (long s = sectors > 2 ? sectors - 2 : 0; s < sectors && !rc; s++) rc = emit(s);
    std::cout.flush();
    if (rc) fprintf(stderr, "read_altb: %s (%s)\n", wrp_strerror(rc), wrp_last_hip_error(eng));
    wrp_destroy(eng);
    return rc ? 1 : 0;
}
