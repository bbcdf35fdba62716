// The set-up record of a LARGE triangle and the per-pixel fragment test of the visibility pass (visibility.hip): rasterSmall queues
// triangles whose bounding box exceeds RT_SMALL_BOX pixels (the ground slab's faces, a model triangle next to the camera), rasterLarge
// takes, for every pixel, the minimum of the pixel's key and the keys of the queued triangles that cover it.
#pragma once
#include "rtggx_device.h"

namespace rt {

struct LargeTri { int32_t X[3], Y[3]; float z[3]; uint32_t word; double invA; uint32_t tl, pad; };   // 56 B; set-up done once, by rasterSmall

// Fragment test + depth for one pixel; returns the key or ~0 when not covered.
// Snapped coordinates are below 2^30 in magnitude (rasterVertex), pixel centres below 2^23: every difference fits 32
// bits, every product is one 32x32->64 multiply-add.
RT_DEV unsigned long long fragmentKey(int32_t PX, int32_t PY, const int32_t X[3], const int32_t Y[3],
                                      bool tl0, bool tl1, bool tl2, double invA, double z0, double dz1, double dz2, uint32_t word) {
  const long long w0 = (long long)(X[2] - X[1]) * (long long)(PY - Y[1]) - (long long)(Y[2] - Y[1]) * (long long)(PX - X[1]);
  const long long w1 = (long long)(X[0] - X[2]) * (long long)(PY - Y[2]) - (long long)(Y[0] - Y[2]) * (long long)(PX - X[2]);
  const long long w2 = (long long)(X[1] - X[0]) * (long long)(PY - Y[0]) - (long long)(Y[1] - Y[0]) * (long long)(PX - X[0]);
  if (w0 < 0 || w1 < 0 || w2 < 0) return ~0ull;
  if ((w0 == 0 && !tl0) || (w1 == 0 && !tl1) || (w2 == 0 && !tl2)) return ~0ull;
  const double l1 = (double)w1 * invA, l2 = (double)w2 * invA;
  const double z = z0 + l1 * dz1 + l2 * dz2;
  if (!(z >= 0.0) || !(z <= 1.0)) return ~0ull;
  const uint32_t d24 = (uint32_t)(z * 16777215.0 + 0.5);
  return ((unsigned long long)d24 << 32) | word;
}

// The keys of the queued large triangles merged into `key` for the pixel (px, py); [tileX0, tileX1] x [tileY0, tileY1]: the pixels of the
// calling wave or workgroup (uniform), whose bounding box decides whether a triangle is looked at at all.
RT_DEV unsigned long long mergeLargeTris(unsigned long long key, uint32_t px, uint32_t py, uint32_t tileX0, uint32_t tileY0, uint32_t tileX1, uint32_t tileY1,
                                         const LargeTri* __restrict__ large, uint32_t n) {
  const int32_t bx0 = (int32_t)tileX0 * 256 + 128, bx1 = (int32_t)tileX1 * 256 + 128, by0 = (int32_t)tileY0 * 256 + 128, by1 = (int32_t)tileY1 * 256 + 128;
  const int32_t PX = (int32_t)px * 256 + 128, PY = (int32_t)py * 256 + 128;
  for (uint32_t i = 0; i < n; ++i) {
    const LargeTri lt = large[i];
    const int32_t minX = min(lt.X[0], min(lt.X[1], lt.X[2])), maxX = max(lt.X[0], max(lt.X[1], lt.X[2]));
    const int32_t minY = min(lt.Y[0], min(lt.Y[1], lt.Y[2])), maxY = max(lt.Y[0], max(lt.Y[1], lt.Y[2]));
    if (maxX < bx0 || minX > bx1 || maxY < by0 || minY > by1) continue;   // uniform
    const double z0 = (double)lt.z[0], dz1 = (double)lt.z[1] - z0, dz2 = (double)lt.z[2] - z0;
    const unsigned long long k = fragmentKey(PX, PY, lt.X, lt.Y, lt.tl & 1u, (lt.tl >> 1) & 1u, (lt.tl >> 2) & 1u, lt.invA, z0, dz1, dz2, lt.word);
    key = k < key ? k : key;
  }
  return key;
}

}  // namespace rt
