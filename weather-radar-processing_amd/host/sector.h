// sector.h -- wire-format container, API-compatible with the reference's Sector (sector.h:7-19).
#ifndef WRP_HOST_SECTOR_H
#define WRP_HOST_SECTOR_H
#include <istream>

class Sector {
  public:
    int sweeps, samples;
    short *hh, *vv, *vh;   // interleaved I,Q: 2*sweeps*samples shorts per channel
    short number;

    Sector(int num_sweeps, int num_samples);
    ~Sector();
    Sector(const Sector &) = delete;
    Sector &operator=(const Sector &) = delete;

    void read(std::istream &in);        // big-endian byte stream, same 12-byte sample layout
    void fromByteArray(char *buff);     // sweeps*samples samples of 12 bytes (sector.cpp:52-62)
    void print() const;
};
#endif
