// framing.h -- result egress framing.
//   rpv2.cu:631-644      : [sector BE16][elevation BE16][gates x BE float]   (ZeroMQ topics "B" = Zdb, "C" = Zdr)
//   read_single.cc:510-517, gpu_1fp_streamcasc.cu:712-719 : [sector BE16][gates x BE float]   (UDP 19002 / 19003)
#ifndef WRP_HOST_FRAMING_H
#define WRP_HOST_FRAMING_H
#include <stddef.h>

// zdb_zdr: [gates][2] as produced by the engine (rpv2.cu:211-212).  which: 0 = Zdb, 1 = Zdr.
// Returns the number of bytes written to out (4*gates + 4 or + 2).
size_t frame_result(const float *zdb_zdr, int gates, int sector, int elevation, int which, bool with_elevation,
                    unsigned char *out);
#endif
