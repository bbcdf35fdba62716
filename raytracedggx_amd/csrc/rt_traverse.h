// Ray set-up and the two intersection tests of the traversal (trace.hip), shared with the kernels that re-derive
// barycentrics from a hit key (raytrace.hip): the same code gives the same bits.
#pragma once
#include "rtggx_device.h"

namespace rt {

struct LaneRay {
  float ox, oy, oz, ix, iy, iz;     // object-space origin, reciprocal direction
  float Sx, Sy, Sz;                 // Woop shear
  int kx, ky, kz;
};

RT_DEV float pick3(float a, float b, float c, int k) { return k == 0 ? a : (k == 1 ? b : c); }

RT_DEV LaneRay toObject(float wox, float woy, float woz, float wdx, float wdy, float wdz, const float* __restrict__ inv) {
  LaneRay r;
  r.ox = ((wox * inv[0] + woy * inv[4]) + woz * inv[8]) + inv[12];
  r.oy = ((wox * inv[1] + woy * inv[5]) + woz * inv[9]) + inv[13];
  r.oz = ((wox * inv[2] + woy * inv[6]) + woz * inv[10]) + inv[14];
  const float dx = (wdx * inv[0] + wdy * inv[4]) + wdz * inv[8];
  const float dy = (wdx * inv[1] + wdy * inv[5]) + wdz * inv[9];
  const float dz = (wdx * inv[2] + wdy * inv[6]) + wdz * inv[10];
  r.ix = 1.0f / dx; r.iy = 1.0f / dy; r.iz = 1.0f / dz;
  const float ax = fabsf(dx), ay = fabsf(dy), az = fabsf(dz);
  r.kz = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);
  r.kx = (r.kz + 1) % 3; r.ky = (r.kx + 1) % 3;
  const float dkz = pick3(dx, dy, dz, r.kz);
  if (dkz < 0.0f) { const int t = r.kx; r.kx = r.ky; r.ky = t; }
  r.Sx = pick3(dx, dy, dz, r.kx) / dkz; r.Sy = pick3(dx, dy, dz, r.ky) / dkz; r.Sz = 1.0f / dkz;
  return r;
}

RT_DEV bool woopTest(const LaneRay& r, const float4 t0, const float4 t1, const float4 t2, float& t, float& b1, float& b2) {
  // t0 = v0.xyz v1.x | t1 = v1.yz v2.xy | t2 = v2.z pad pad pad
  const float Ax0 = t0.x - r.ox, Ay0 = t0.y - r.oy, Az0 = t0.z - r.oz;
  const float Bx0 = t0.w - r.ox, By0 = t1.x - r.oy, Bz0 = t1.y - r.oz;
  const float Cx0 = t1.z - r.ox, Cy0 = t1.w - r.oy, Cz0 = t2.x - r.oz;
  const float Akz = pick3(Ax0, Ay0, Az0, r.kz), Bkz = pick3(Bx0, By0, Bz0, r.kz), Ckz = pick3(Cx0, Cy0, Cz0, r.kz);
  const float Ax = pick3(Ax0, Ay0, Az0, r.kx) - r.Sx * Akz, Ay = pick3(Ax0, Ay0, Az0, r.ky) - r.Sy * Akz;
  const float Bx = pick3(Bx0, By0, Bz0, r.kx) - r.Sx * Bkz, By = pick3(Bx0, By0, Bz0, r.ky) - r.Sy * Bkz;
  const float Cx = pick3(Cx0, Cy0, Cz0, r.kx) - r.Sx * Ckz, Cy = pick3(Cx0, Cy0, Cz0, r.ky) - r.Sy * Ckz;
  float U = Cx * By - Cy * Bx, V = Ax * Cy - Ay * Cx, W = Bx * Ay - By * Ax;
  if (U == 0.0f || V == 0.0f || W == 0.0f) {
    U = (float)((double)Cx * (double)By - (double)Cy * (double)Bx);
    V = (float)((double)Ax * (double)Cy - (double)Ay * (double)Cx);
    W = (float)((double)Bx * (double)Ay - (double)By * (double)Ax);
  }
  if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return false;
  const float det = (U + V) + W;
  if (det == 0.0f) return false;
  const float Az = r.Sz * Akz, Bz = r.Sz * Bkz, Cz = r.Sz * Ckz;
  const float T = (U * Az + V * Bz) + W * Cz;
  const float rdet = 1.0f / det;
  t = T * rdet; b1 = V * rdet; b2 = W * rdet;
  return true;
}

RT_DEV void slabTest(const LaneRay& r, float mnx, float mny, float mnz, float mxx, float mxy, float mxz, float tmin, float tmax, float& tn, float& tf) {
  const float x1 = (mnx - r.ox) * r.ix, x2 = (mxx - r.ox) * r.ix;
  const float y1 = (mny - r.oy) * r.iy, y2 = (mxy - r.oy) * r.iy;
  const float z1 = (mnz - r.oz) * r.iz, z2 = (mxz - r.oz) * r.iz;
  tn = fmaxf(fmaxf(fminf(x1, x2), fminf(y1, y2)), fmaxf(fminf(z1, z2), tmin));
  tf = fminf(fminf(fmaxf(x1, x2), fmaxf(y1, y2)), fminf(fmaxf(z1, z2), tmax));
}

// The barycentrics (and t) of `r` against one triangle, as the traversal computed them when it recorded the hit.
RT_DEV bool woopTestVerts(const LaneRay& r, f3 v0, f3 v1, f3 v2, float& t, float& b1, float& b2) {
  return woopTest(r, make_float4(v0.x, v0.y, v0.z, v1.x), make_float4(v1.y, v1.z, v2.x, v2.y), make_float4(v2.z, 0.0f, 0.0f, 0.0f), t, b1, b2);
}

}  // namespace rt
