// rpv2.cpp -- entry point with the reference's name (Makefile:4 builds `rpv2`; main.cpp:3-16 is
// the class-based variant).  usage:
//   rpv2 [num_streams] [--in udp:PORT | file:PATH] [--out udp:PORT_ZDB,PORT_ZDR | file:PATH | none]
//        [--sectors N] [--device D] [--no-elevation]
// Defaults reproduce main.cpp:10-15: 143 sectors x 9 elevations of 1024 x 512, UDP 19001 in,
// 19002 / 19003 out.  file: input is a concatenation of wire-format sectors (12 bytes/sample),
// file: output a concatenation of frames, Zdb then Zdr per sector.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>

#include "radar_processor.h"

int main(int argc, char **argv)
{
    int num_streams = 2, device = 0;
    long sectors = -1;
    bool with_elev = true, with_elev_set = false;
    std::string in = "udp:19001", out = "udp:19002,19003";
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a == "--in" && i + 1 < argc) in = argv[++i];
        else if (a == "--out" && i + 1 < argc) out = argv[++i];
        else if (a == "--sectors" && i + 1 < argc) sectors = atol(argv[++i]);
        else if (a == "--device" && i + 1 < argc) device = atoi(argv[++i]);
        else if (a == "--no-elevation") { with_elev = false; with_elev_set = true; }
        else if (a[0] != '-') { num_streams = atoi(a.c_str()); if (num_streams < 1) num_streams = 1; }
        else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    RadarProcessor proc(143, 1024, 512, 9, num_streams);
    proc.set_device(device);
    proc.set_max_sectors(sectors);
    FILE *fin = nullptr, *fout = nullptr;
    try {
        if (in.rfind("udp:", 0) == 0 && out.rfind("udp:", 0) == 0) {
            int ports[2] = {19002, 19003};
            sscanf(out.c_str() + 4, "%d,%d", &ports[0], &ports[1]);
            proc.set_comms(atoi(in.c_str() + 4), ports, 2);
        } else {
            if (in.rfind("file:", 0) == 0) {
                fin = fopen(in.c_str() + 5, "rb");
                if (!fin) { perror("input"); return 2; }
                proc.set_source([fin](char *buf, size_t bytes) { return fread(buf, 1, bytes, fin) == bytes; });
            } else { fprintf(stderr, "mixing udp and file endpoints is not supported\n"); return 2; }
            if (out.rfind("file:", 0) == 0) {
                fout = fopen(out.c_str() + 5, "wb");
                if (!fout) { perror("output"); return 2; }
                proc.set_sink([fout](int, int, int, const unsigned char *f, size_t n) { fwrite(f, 1, n, fout); });
            }
        }
    } catch (const char *msg) {
        fprintf(stderr, "socket: %s\n", msg);
        return 3;
    }
    if (with_elev_set) proc.set_frame_with_elevation(with_elev);
    const int rc = proc.start();
    if (fin) fclose(fin);
    if (fout) fclose(fout);
    if (rc) fprintf(stderr, "rpv2: %s\n", proc.last_error());
    else fprintf(stderr, "rpv2: %ld sectors processed\n", proc.sectors_done());
    return rc ? 1 : 0;
}
