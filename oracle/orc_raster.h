// ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked, imported or called by the product path.
// Visibility pass of the reference, restated for the CPU:
//   RayTracer::visibility            RayTracedGGX/Content/RayTracer.cpp:751-791 (clear, 2x DrawIndexed)
//   VSVisibility.hlsl:26-32          pos = mul(float4(Pos,1), WVP); pos.xy += bias * pos.w
//   PSVisibility.hlsl:18-24          ((instance << 24) | primitiveId) + 1
// The rasteriser itself is Direct3D's (not in the tree).  Rules restated from the D3D11
// functional spec: viewport transform, 8 sub-pixel bits of snapping, pixel centres at +0.5,
// top-left fill rule, back-face culling with clockwise = front (XUSG preset CULL_BACK,
// RayTracedGGX/XUSG/Core/XUSG.h:2302-2319), depth test LESS against a D24_UNORM buffer cleared
// to 1.0, depth clip to [0,1].  Choices where the spec leaves latitude:
//   * z is interpolated with barycentrics from the snapped integer edge functions, in double;
//   * D24 = floor(z * (2^24-1) + 0.5);
//   * near plane: a triangle with vertices on both sides of z_clip = 0 is clipped (Sutherland-Hodgman, one plane,
//     new vertices by fp32 interpolation in clip space from the inside vertex towards the outside one, z set to 0)
//     into one or two triangles that share the primitive id; the far plane needs no geometry (z <= 1 per pixel);
//   * guard band (round 4): a polygon (after the near clip) with a vertex outside |x| <= 256 w or |y| <= 256 w is clipped against those
//     four planes, in that order (x <= 256 w, x >= -256 w, y <= 256 w, y >= -256 w), Sutherland-Hodgman in clip space, fp32, new
//     vertices interpolated from the inside vertex towards the outside one with the clipped coordinate set to +-256 w, and the result
//     is rasterised as a fan around its first vertex: snapped coordinates then stay below 2^30 for every frame size the context accepts
//     (256 + 1 viewport half-widths of at most 8192 pixels x 256 sub-pixels = 2^29).  D3D's rasteriser clips to a guard band of its own
//     choosing; which one is not observable from the reference, so this is the contract;
//   * a (sub-)triangle with a vertex at w <= 0 is dropped.
// Equal depth keeps the earlier fragment (LESS): lower instance first, then lower primitive.
#pragma once
#include "orc_scene.h"

namespace orc {

struct RasterVert { int64_t X, Y; float z; bool ok; };

// clip-space position of a vertex (VSVisibility.hlsl:26-32)
static inline float4 clip_vertex(const float* pos, const M4& wvp, const float bias[2]) {
  float4 p = mul_point(f3(pos[0], pos[1], pos[2]), wvp);
  p.x += bias[0] * p.w;
  p.y += bias[1] * p.w;
  return p;
}
// viewport transform + snapping to 8 sub-pixel bits
static inline RasterVert raster_vertex(const float4& p, uint32_t W, uint32_t H) {
  RasterVert r{0, 0, 0.0f, false};
  if (!(p.w > 0.0f)) return r;
  const float nx = p.x / p.w, ny = p.y / p.w;
  r.z = p.z / p.w;
  const float sx = (nx + 1.0f) * ((float)W * 0.5f);
  const float sy = (1.0f - ny) * ((float)H * 0.5f);
  const float fx = std::floor(sx * 256.0f + 0.5f), fy = std::floor(sy * 256.0f + 0.5f);
  if (!(std::fabs(fx) < 1073741824.0f) || !(std::fabs(fy) < 1073741824.0f)) return r;
  r.X = (int64_t)fx; r.Y = (int64_t)fy; r.ok = true;
  return r;
}
// Near-plane clip of one triangle: writes 0, 3 or 4 vertices (a convex polygon in the input's winding) and returns
// the count.  A vertex with z >= 0 is inside.
static inline int clip_near(const float4 in[3], float4 out[4]) {
  int n = 0;
  for (int k = 0; k < 3; ++k) {
    const float4& a = in[k]; const float4& b = in[(k + 1) % 3];
    const bool ia = a.z >= 0.0f, ib = b.z >= 0.0f;
    if (ia) out[n++] = a;
    if (ia != ib) {
      const float4& p = ia ? a : b; const float4& q = ia ? b : a;       // from the inside vertex to the outside one
      const float t = p.z / (p.z - q.z);
      float4 c; c.x = p.x + (q.x - p.x) * t; c.y = p.y + (q.y - p.y) * t; c.z = 0.0f; c.w = p.w + (q.w - p.w) * t;
      out[n++] = c;
    }
  }
  return n;
}

// Guard-band clip of a convex polygon against plane 1..4 (see the header); returns the new vertex count (at most one more).
static const float kGuard = 256.0f;
static inline float guard_distance(const float4& v, int plane) {
  const float gw = kGuard * v.w;
  return plane == 1 ? gw - v.x : plane == 2 ? gw + v.x : plane == 3 ? gw - v.y : gw + v.y;
}
static inline int clip_guard(const float4* in, int n, float4* out, int plane) {
  int m = 0;
  for (int k = 0; k < n; ++k) {
    const float4& a = in[k]; const float4& b = in[(k + 1) % n];
    const float da = guard_distance(a, plane), db = guard_distance(b, plane);
    const bool ia = da >= 0.0f, ib = db >= 0.0f;
    if (ia) out[m++] = a;
    if (ia != ib) {
      const float4& p = ia ? a : b; const float4& q = ia ? b : a;
      const float dp = ia ? da : db, dq = ia ? db : da;
      const float t = dp / (dp - dq);
      float4 c; c.x = p.x + (q.x - p.x) * t; c.y = p.y + (q.y - p.y) * t; c.z = p.z + (q.z - p.z) * t; c.w = p.w + (q.w - p.w) * t;
      const float gw = kGuard * c.w;
      if (plane == 1) c.x = gw; else if (plane == 2) c.x = -gw; else if (plane == 3) c.y = gw; else c.y = -gw;
      out[m++] = c;
    }
  }
  return m;
}
static inline bool outside_guard(const float4& v) { const float gw = kGuard * v.w; return std::fabs(v.x) > gw || std::fabs(v.y) > gw; }

static inline bool is_top_left(int64_t ax, int64_t ay, int64_t bx, int64_t by) {
  // front faces are clockwise in y-down screen space (area2 > 0): a top edge runs left->right
  // on a horizontal line, a left edge runs upwards.
  const int64_t dx = bx - ax, dy = by - ay;
  return (dy == 0 && dx > 0) || dy < 0;
}

static inline void render_visibility(Ctx& c) {
  const uint32_t W = c.W, H = c.H;
  std::vector<uint64_t> key((size_t)W * H, ((uint64_t)0xFFFFFFu << 32));   // depth 1.0, visibility 0
  for (uint32_t inst = 0; inst < 2; ++inst) {
    const Mesh& m = c.mesh[inst];
    const M4 wvp = cb_load4x4(c.fc.po[inst].WorldViewProj);
    const float* bias = c.fc.po[inst].ProjBias;
    const uint32_t ntri = (uint32_t)(m.idx.size() / 3);
    for (uint32_t prim = 0; prim < ntri; ++prim) {
      float4 cp[3], poly[8], tmp[8];
      for (int k = 0; k < 3; ++k) cp[k] = clip_vertex(&m.verts[6 * (size_t)m.idx[3 * prim + k]], wvp, bias);
      const bool allIn = cp[0].z >= 0.0f && cp[1].z >= 0.0f && cp[2].z >= 0.0f;
      int nv = 3;
      if (allIn) { poly[0] = cp[0]; poly[1] = cp[1]; poly[2] = cp[2]; } else nv = clip_near(cp, poly);
      bool guard = false;
      for (int k = 0; k < nv; ++k) guard = guard || outside_guard(poly[k]);
      if (guard) {
        nv = clip_guard(poly, nv, tmp, 1); nv = clip_guard(tmp, nv, poly, 2);
        nv = clip_guard(poly, nv, tmp, 3); nv = clip_guard(tmp, nv, poly, 4);
      }
      for (int sub = 0; sub + 2 < nv; ++sub) {          // fan: (0,1,2), (0,2,3)
      RasterVert v[3];
      v[0] = raster_vertex(poly[0], W, H); v[1] = raster_vertex(poly[sub + 1], W, H); v[2] = raster_vertex(poly[sub + 2], W, H);
      if (!(v[0].ok && v[1].ok && v[2].ok)) continue;
      const int64_t area2 = (v[1].X - v[0].X) * (v[2].Y - v[0].Y) - (v[1].Y - v[0].Y) * (v[2].X - v[0].X);
      if (area2 <= 0) continue;                                         // back-facing or degenerate
      int64_t minX = std::min(v[0].X, std::min(v[1].X, v[2].X)), maxX = std::max(v[0].X, std::max(v[1].X, v[2].X));
      int64_t minY = std::min(v[0].Y, std::min(v[1].Y, v[2].Y)), maxY = std::max(v[0].Y, std::max(v[1].Y, v[2].Y));
      // pixel px is a candidate when its centre px*256+128 lies in [min, max]
      int64_t px0 = (minX - 128 + 255) >> 8, px1 = (maxX - 128) >> 8;
      int64_t py0 = (minY - 128 + 255) >> 8, py1 = (maxY - 128) >> 8;
      px0 = std::max<int64_t>(px0, 0); py0 = std::max<int64_t>(py0, 0);
      px1 = std::min<int64_t>(px1, (int64_t)W - 1); py1 = std::min<int64_t>(py1, (int64_t)H - 1);
      if (px0 > px1 || py0 > py1) continue;
      const bool tl0 = is_top_left(v[1].X, v[1].Y, v[2].X, v[2].Y);
      const bool tl1 = is_top_left(v[2].X, v[2].Y, v[0].X, v[0].Y);
      const bool tl2 = is_top_left(v[0].X, v[0].Y, v[1].X, v[1].Y);
      const double invA = 1.0 / (double)area2;
      const double z0 = (double)v[0].z, dz1 = (double)v[1].z - z0, dz2 = (double)v[2].z - z0;
      const uint32_t word = ((inst << 24) | prim) + 1u;
      for (int64_t py = py0; py <= py1; ++py) for (int64_t px = px0; px <= px1; ++px) {
        const int64_t PX = px * 256 + 128, PY = py * 256 + 128;
        const int64_t w0 = (v[2].X - v[1].X) * (PY - v[1].Y) - (v[2].Y - v[1].Y) * (PX - v[1].X);
        const int64_t w1 = (v[0].X - v[2].X) * (PY - v[2].Y) - (v[0].Y - v[2].Y) * (PX - v[2].X);
        const int64_t w2 = (v[1].X - v[0].X) * (PY - v[0].Y) - (v[1].Y - v[0].Y) * (PX - v[0].X);
        if (w0 < 0 || w1 < 0 || w2 < 0) continue;
        if ((w0 == 0 && !tl0) || (w1 == 0 && !tl1) || (w2 == 0 && !tl2)) continue;
        const double l1 = (double)w1 * invA, l2 = (double)w2 * invA;
        const double z = z0 + l1 * dz1 + l2 * dz2;
        if (!(z >= 0.0) || !(z <= 1.0)) continue;                       // depth clip
        const uint32_t d24 = (uint32_t)(z * 16777215.0 + 0.5);
        const uint64_t k = ((uint64_t)d24 << 32) | word;
        uint64_t& dst = key[(size_t)py * W + (size_t)px];
        if (k < dst) dst = k;
      }
      }   // sub
    }
  }
  c.vis.resize((size_t)W * H); c.depth.resize((size_t)W * H);
  for (size_t i = 0; i < (size_t)W * H; ++i) { c.vis[i] = (uint32_t)key[i]; c.depth[i] = (uint32_t)(key[i] >> 32); }
}

}  // namespace orc
