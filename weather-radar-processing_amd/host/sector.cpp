// sector.cpp -- Sector: decode of the radar's wire format.
// Format (reference sector.cpp:52-62): per sample 12 bytes hhI hhQ vvI vvQ vhI vhQ, each a
// big-endian int16.  Written from scratch: one pointer walk with a defined evaluation order
// (the reference relies on unsequenced idx++ pairs).
#include "sector.h"

#include <iostream>

namespace {
inline short be16(const unsigned char *p) { return (short)(((unsigned)p[0] << 8) | (unsigned)p[1]); }
}

Sector::Sector(int num_sweeps, int num_samples) : sweeps(num_sweeps), samples(num_samples), number(0)
{
    const size_t n = 2 * (size_t)sweeps * samples;
    hh = new short[n];
    vv = new short[n];
    vh = new short[n];
}

Sector::~Sector()
{
    delete[] hh;
    delete[] vv;
    delete[] vh;
}

void Sector::fromByteArray(char *buff)
{
    const unsigned char *p = reinterpret_cast<const unsigned char *>(buff);
    const size_t count = (size_t)sweeps * samples;
    for (size_t i = 0; i < count; i++, p += 12) {
        hh[2 * i] = be16(p);
        hh[2 * i + 1] = be16(p + 2);
        vv[2 * i] = be16(p + 4);
        vv[2 * i + 1] = be16(p + 6);
        vh[2 * i] = be16(p + 8);
        vh[2 * i + 1] = be16(p + 10);
    }
}

void Sector::read(std::istream &in)
{
    // same layout, from a stream; stops at end of data or when the sector is full
    const size_t count = (size_t)sweeps * samples;
    unsigned char b[12];
    for (size_t i = 0; i < count; i++) {
        in.read(reinterpret_cast<char *>(b), 12);
        if (in.gcount() != 12) break;
        hh[2 * i] = be16(b);
        hh[2 * i + 1] = be16(b + 2);
        vv[2 * i] = be16(b + 4);
        vv[2 * i + 1] = be16(b + 6);
        vh[2 * i] = be16(b + 8);
        vh[2 * i + 1] = be16(b + 10);
    }
}

void Sector::print() const
{
    const size_t count = (size_t)sweeps * samples;
    const short *ch[3] = {hh, vv, vh};
    const char *name[3] = {"hh:", "vv:", "vh:"};
    for (int c = 0; c < 3; c++) {
        std::cout << name[c] << std::endl;
        for (size_t i = 0; i < count; i++) std::cout << ch[c][2 * i] << " " << ch[c][2 * i + 1] << " ";
        std::cout << std::endl;
    }
}
