// dimension.h -- buffer index maps, API-compatible with the reference (dimension.h:4-16).
#ifndef WRP_HOST_DIMENSION_H
#define WRP_HOST_DIMENSION_H

class Dimension3 {
  public:
    const int width, height, depth, m_size, total_size;
    int at_depth(int x, int y, int depth);
    Dimension3(int w, int h, int d);
};

// [depth = stream slot][copy = channel][y = row][x = column]
class Dimension4 {
  public:
    const int width, height, copies, depth, m_size, total_size;
    int copy_at_depth(int x, int y, int copy, int depth);
    Dimension4(int w, int h, int c, int d);
};
#endif
