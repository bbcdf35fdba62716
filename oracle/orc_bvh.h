// ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked, imported or called by the product path.
// Closest-hit ray query over the two-instance scene: the CPU stand-in for DXR's
// TraceRay(g_scene, RAY_FLAG_NONE, ~0, hitGroup, 1, 0, ray, payload)  (RayTracing.hlsl:183-198)
// and for the driver-built BLAS/TLAS (RayTracer.cpp:676-716, 158-233, 326-341).  The acceleration
// structure and the ray/triangle test live in the D3D12 driver and hardware: "parity unpinned"
// (SURVEY.md 8c); only closest-hit semantics are observable.  Definitions fixed here:
//   * two-level: rays are carried into each instance's object space with the world->object
//     matrix, t is preserved; instance 0 (ground) is searched before instance 1 (model);
//   * intersection test: watertight test of Woop, Benthin, Wald (JCGT 2013) without back-face
//     culling, double-precision fallback when an edge function is exactly zero -- DXR requires
//     watertight intersection;
//   * a hit needs TMin < t < TMax (exclusive, DXR triangle rule); equal t keeps the lower
//     (instance, primitive);
//   * the (instance, primitive) a depth-0 ray starts on can be excluded (SURVEY.md App. A,
//     "Self-intersection": neutral in exact arithmetic);
//   * traversal: binary BVH, both child boxes in the parent, nearer child first.  The result is
//     meant to be independent of the BVH (tests check it against the brute-force loop below).
#pragma once
#include <algorithm>
#include <cfloat>
#include "orc_scene.h"

namespace orc {

struct Hit { float t; uint32_t inst, prim; float b1, b2; bool valid; };

struct RayXform {      // per-instance object-space ray + Woop shear constants
  float o[3], d[3], invd[3];
  int kx, ky, kz; float Sx, Sy, Sz;
};

static inline RayXform ray_to_object(const float3 o, const float3 d, const M4& inv) {
  RayXform r;
  const float4 oo = mul_point(o, inv);
  const float3 dd = mul_dir(d, inv);
  r.o[0] = oo.x; r.o[1] = oo.y; r.o[2] = oo.z;
  r.d[0] = dd.x; r.d[1] = dd.y; r.d[2] = dd.z;
  for (int k = 0; k < 3; ++k) r.invd[k] = 1.0f / r.d[k];
  const float ax = std::fabs(r.d[0]), ay = std::fabs(r.d[1]), az = std::fabs(r.d[2]);
  r.kz = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);
  r.kx = (r.kz + 1) % 3; r.ky = (r.kx + 1) % 3;
  if (r.d[r.kz] < 0.0f) std::swap(r.kx, r.ky);
  r.Sx = r.d[r.kx] / r.d[r.kz]; r.Sy = r.d[r.ky] / r.d[r.kz]; r.Sz = 1.0f / r.d[r.kz];
  return r;
}

// Watertight ray/triangle test; returns true and (t, b1, b2) when the ray's line meets the triangle.
static inline bool woop_intersect(const RayXform& r, const BvhTri& tr, float& t, float& b1, float& b2) {
  const float A[3] = {tr.v0[0] - r.o[0], tr.v0[1] - r.o[1], tr.v0[2] - r.o[2]};
  const float B[3] = {tr.v1[0] - r.o[0], tr.v1[1] - r.o[1], tr.v1[2] - r.o[2]};
  const float C[3] = {tr.v2[0] - r.o[0], tr.v2[1] - r.o[1], tr.v2[2] - r.o[2]};
  const float Ax = A[r.kx] - r.Sx * A[r.kz], Ay = A[r.ky] - r.Sy * A[r.kz];
  const float Bx = B[r.kx] - r.Sx * B[r.kz], By = B[r.ky] - r.Sy * B[r.kz];
  const float Cx = C[r.kx] - r.Sx * C[r.kz], Cy = C[r.ky] - r.Sy * C[r.kz];
  float U = Cx * By - Cy * Bx, V = Ax * Cy - Ay * Cx, W = Bx * Ay - By * Ax;
  if (U == 0.0f || V == 0.0f || W == 0.0f) {
    U = (float)((double)Cx * (double)By - (double)Cy * (double)Bx);
    V = (float)((double)Ax * (double)Cy - (double)Ay * (double)Cx);
    W = (float)((double)Bx * (double)Ay - (double)By * (double)Ax);
  }
  if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return false;
  const float det = (U + V) + W;
  if (det == 0.0f) return false;
  const float Az = r.Sz * A[r.kz], Bz = r.Sz * B[r.kz], Cz = r.Sz * C[r.kz];
  const float T = (U * Az + V * Bz) + W * Cz;
  const float rdet = 1.0f / det;
  t = T * rdet; b1 = V * rdet; b2 = W * rdet;
  return true;
}

static inline void consider(Hit& best, float tmin, uint32_t inst, const BvhTri& tr, const RayXform& r,
                            uint32_t skipInst, uint32_t skipPrim) {
  if (inst == skipInst && tr.prim == skipPrim) return;
  float t, b1, b2;
  if (!woop_intersect(r, tr, t, b1, b2)) return;
  if (!(t > tmin)) return;
  const bool closer = t < best.t;
  const bool tie = best.valid && t == best.t && (inst < best.inst || (inst == best.inst && tr.prim < best.prim));
  if (closer || tie) { best.t = t; best.inst = inst; best.prim = tr.prim; best.b1 = b1; best.b2 = b2; best.valid = true; }
}

static inline void slab(const RayXform& r, const float* bmin, const float* bmax, float tmin, float tmax, float& tn, float& tf) {
  const float x1 = (bmin[0] - r.o[0]) * r.invd[0], x2 = (bmax[0] - r.o[0]) * r.invd[0];
  const float y1 = (bmin[1] - r.o[1]) * r.invd[1], y2 = (bmax[1] - r.o[1]) * r.invd[1];
  const float z1 = (bmin[2] - r.o[2]) * r.invd[2], z2 = (bmax[2] - r.o[2]) * r.invd[2];
  tn = std::fmax(std::fmax(std::fmin(x1, x2), std::fmin(y1, y2)), std::fmax(std::fmin(z1, z2), tmin));
  tf = std::fmin(std::fmin(std::fmax(x1, x2), std::fmax(y1, y2)), std::fmin(std::fmax(z1, z2), tmax));
}

// traversal statistics for performance analysis (node visits, leaf tests, deepest stack), per thread
struct TraverseStats { uint64_t nodes = 0, leaves = 0, rays = 0; int maxStack = 0; };
static thread_local TraverseStats g_tstats;

static inline void traverse_blas(const Bvh& bvh, const RayXform& r, float tmin, uint32_t inst, Hit& best,
                                 uint32_t skipInst, uint32_t skipPrim) {
  if (bvh.tris.empty()) return;
  int32_t stack[64]; int sp = 0;
  int32_t cur = bvh.root;
  for (;;) {
    if (cur < 0) {
      ++g_tstats.leaves;
      consider(best, tmin, inst, bvh.tris[(size_t)~cur], r, skipInst, skipPrim);
      if (sp == 0) break;
      cur = stack[--sp];
      continue;
    }
    const BvhNode& n = bvh.nodes[(size_t)cur];
    ++g_tstats.nodes;
    if (sp > g_tstats.maxStack) g_tstats.maxStack = sp;
    float ln, lf, rn, rf;
    slab(r, n.lmin, n.lmax, tmin, best.t, ln, lf);
    slab(r, n.rmin, n.rmax, tmin, best.t, rn, rf);
    const bool hl = ln <= lf * 1.0000004f, hr = rn <= rf * 1.0000004f;
    if (hl && hr) {
      if (ln <= rn) { stack[sp++] = n.right; cur = n.left; } else { stack[sp++] = n.left; cur = n.right; }
    } else if (hl) cur = n.left;
    else if (hr) cur = n.right;
    else { if (sp == 0) break; cur = stack[--sp]; }
  }
}

static inline Hit trace_closest(const Ctx& c, float3 o, float3 d, float tmin, float tmax,
                                uint32_t skipInst = 0xFFFFFFFFu, uint32_t skipPrim = 0xFFFFFFFFu) {
  Hit best{tmax, 0, 0, 0, 0, false};
  if (!(tmax > tmin)) return best;
  for (uint32_t inst = 0; inst < 2; ++inst) {
    const RayXform r = ray_to_object(o, d, c.invWorld[inst]);
    traverse_blas(c.mesh[inst].bvh, r, tmin, inst, best, skipInst, skipPrim);
  }
  return best;
}

// Reference semantics without any acceleration structure (for tests of BVH independence).
static inline Hit trace_brute(const Ctx& c, float3 o, float3 d, float tmin, float tmax,
                              uint32_t skipInst = 0xFFFFFFFFu, uint32_t skipPrim = 0xFFFFFFFFu) {
  Hit best{tmax, 0, 0, 0, 0, false};
  if (!(tmax > tmin)) return best;
  for (uint32_t inst = 0; inst < 2; ++inst) {
    const RayXform r = ray_to_object(o, d, c.invWorld[inst]);
    const Mesh& m = c.mesh[inst];
    const uint32_t ntri = (uint32_t)(m.idx.size() / 3);
    for (uint32_t p = 0; p < ntri; ++p) {
      BvhTri tr;
      for (int k = 0; k < 3; ++k) { tr.v0[k] = m.verts[6 * (size_t)m.idx[3 * p] + k]; tr.v1[k] = m.verts[6 * (size_t)m.idx[3 * p + 1] + k]; tr.v2[k] = m.verts[6 * (size_t)m.idx[3 * p + 2] + k]; }
      tr.prim = p;
      consider(best, tmin, inst, tr, r, skipInst, skipPrim);
    }
  }
  return best;
}

// The oracle's own builder: recursive median split on the longest centroid axis.  Independent of
// the product's LBVH; used to show that hits do not depend on the hierarchy.
namespace bvhbuild {
struct Ref { float c[3]; float mn[3], mx[3]; uint32_t prim; };
static inline void bounds(const std::vector<Ref>& refs, size_t b, size_t e, float* mn, float* mx) {
  for (int k = 0; k < 3; ++k) { mn[k] = FLT_MAX; mx[k] = -FLT_MAX; }
  for (size_t i = b; i < e; ++i) for (int k = 0; k < 3; ++k) { mn[k] = std::fmin(mn[k], refs[i].mn[k]); mx[k] = std::fmax(mx[k], refs[i].mx[k]); }
}
static inline int32_t build(Bvh& bvh, std::vector<Ref>& refs, size_t b, size_t e, const Mesh& m) {
  if (e - b == 1) {
    BvhTri tr{}; const uint32_t p = refs[b].prim;
    for (int k = 0; k < 3; ++k) { tr.v0[k] = m.verts[6 * (size_t)m.idx[3 * p] + k]; tr.v1[k] = m.verts[6 * (size_t)m.idx[3 * p + 1] + k]; tr.v2[k] = m.verts[6 * (size_t)m.idx[3 * p + 2] + k]; }
    tr.prim = p;
    bvh.tris.push_back(tr);
    return ~(int32_t)(bvh.tris.size() - 1);
  }
  float cmn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, cmx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (size_t i = b; i < e; ++i) for (int k = 0; k < 3; ++k) { cmn[k] = std::fmin(cmn[k], refs[i].c[k]); cmx[k] = std::fmax(cmx[k], refs[i].c[k]); }
  int axis = 0; float ext = cmx[0] - cmn[0];
  for (int k = 1; k < 3; ++k) if (cmx[k] - cmn[k] > ext) { ext = cmx[k] - cmn[k]; axis = k; }
  const size_t mid = (b + e) / 2;
  std::nth_element(refs.begin() + (long)b, refs.begin() + (long)mid, refs.begin() + (long)e,
                   [axis](const Ref& x, const Ref& y) { return x.c[axis] < y.c[axis] || (x.c[axis] == y.c[axis] && x.prim < y.prim); });
  const int32_t id = (int32_t)bvh.nodes.size();
  bvh.nodes.push_back(BvhNode{});
  float lmn[3], lmx[3], rmn[3], rmx[3];
  bounds(refs, b, mid, lmn, lmx); bounds(refs, mid, e, rmn, rmx);
  const int32_t l = build(bvh, refs, b, mid, m);
  const int32_t r = build(bvh, refs, mid, e, m);
  BvhNode& n = bvh.nodes[(size_t)id];
  for (int k = 0; k < 3; ++k) { n.lmin[k] = lmn[k]; n.lmax[k] = lmx[k]; n.rmin[k] = rmn[k]; n.rmax[k] = rmx[k]; }
  n.left = l; n.right = r; n.pad[0] = n.pad[1] = 0;
  return id;
}
}  // namespace bvhbuild

static inline void build_bvh(Mesh& m) {
  using namespace bvhbuild;
  m.bvh = Bvh{};
  const uint32_t ntri = (uint32_t)(m.idx.size() / 3);
  std::vector<Ref> refs(ntri);
  for (uint32_t p = 0; p < ntri; ++p) {
    Ref& r = refs[p]; r.prim = p;
    for (int k = 0; k < 3; ++k) {
      const float a = m.verts[6 * (size_t)m.idx[3 * p] + k], b = m.verts[6 * (size_t)m.idx[3 * p + 1] + k], c = m.verts[6 * (size_t)m.idx[3 * p + 2] + k];
      r.mn[k] = std::fmin(a, std::fmin(b, c)); r.mx[k] = std::fmax(a, std::fmax(b, c));
      r.c[k] = 0.5f * (r.mn[k] + r.mx[k]);
    }
  }
  if (ntri) m.bvh.root = build(m.bvh, refs, 0, ntri, m);
}

}  // namespace orc
