// dimension.h -- row-major index maps for the engine's buffers.
//
// The engine's blocks are dense row-major arrays whose extents are named, innermost first,
// width (column / pulse j), height (row / range cell i), copies (polarisation plane) and depth
// (stream slot).  Dimension3 and Dimension4 keep the class names, public members and method
// names a caller of the reference's dimension.h:4-16 uses; both are thin views over one
// stride calculator so the arithmetic exists once (header-only; dimension.cpp only anchors the
// object file the reference's Makefile lists).
#ifndef WRP_HOST_DIMENSION_H
#define WRP_HOST_DIMENSION_H

namespace wrp_host_detail {
// strides of a dense row-major box with up to four extents (innermost first)
struct Strides {
    int s1, s2, s3;   // elements to step one row, one plane, one slot
    Strides(int e0, int e1, int e2) : s1(e0), s2(e0 * e1), s3(e0 * e1 * e2) {}
    int offset(int i0, int i1, int i2, int i3) const { return i0 + i1 * s1 + i2 * s2 + i3 * s3; }
};
}   // namespace wrp_host_detail

// [depth][height][width]
class Dimension3 {
    wrp_host_detail::Strides strides_;
  public:
    const int width, height, depth;
    const int m_size;       // elements of one width x height matrix
    const int total_size;   // elements of the whole box
    Dimension3(int w, int h, int d)
        : strides_(w, h, 1), width(w), height(h), depth(d), m_size(strides_.s2), total_size(strides_.s2 * d) {}
    int at_depth(int x, int y, int slot) { return strides_.offset(x, y, slot, 0); }
};

// [depth = stream slot][copies = channel][height = row][width = column]; this is the layout of
// the engine's pinned IQ slots and of its result table (rpv2.cu:734-736).
class Dimension4 {
    wrp_host_detail::Strides strides_;
  public:
    const int width, height, copies, depth;
    const int m_size;
    const int total_size;
    Dimension4(int w, int h, int c, int d)
        : strides_(w, h, c), width(w), height(h), copies(c), depth(d), m_size(strides_.s2), total_size(strides_.s3 * d) {}
    int copy_at_depth(int x, int y, int plane, int slot) { return strides_.offset(x, y, plane, slot); }
};

#endif   // WRP_HOST_DIMENSION_H
