// Visibility pass: RayTracer::visibility (RayTracedGGX/Content/RayTracer.cpp:751-791) with
// VSVisibility.hlsl:26-32 and PSVisibility.hlsl:18-24, as a software rasteriser for gfx950
// (CDNA4 has no fixed-function raster).
//
// D3D11 rasterisation rules, integer-exact: 8 sub-pixel bits, pixel centres at +0.5, top-left
// rule, clockwise = front, back faces culled, depth LESS on D24, depth clip [0,1].
// Depth order is resolved with ONE 64-bit atomicMin per fragment on the key
//     (D24 << 32) | (((instance << 24) | primitive) + 1)
// so equal depth keeps the fragment drawn first (lower instance, then lower primitive), exactly
// what LESS does for two ordered draws.  The cleared key is (0xFFFFFF << 32) | 0.
//
// Two kernels: rasterSmall -- one lane per triangle, walks its own bounding box when it covers
// <= 256 pixels (almost every model triangle), otherwise appends a setup record to a queue;
// rasterLarge -- one 64x4 pixel tile per workgroup, one pixel per lane, loops over the queued
// large triangles (the ground slab's faces) and merges with a plain read-min-write.
// Roofline: HBM; algorithmic bytes 8 B/pixel (clear) + 8 B/covered fragment.
#include <hip/hip_ext.h>
#include "rtggx_context.h"
#include "rt_raster.h"

namespace rt {


struct RVert { long long X, Y; float z; bool ok; };

// clip-space position of a vertex (VSVisibility.hlsl:26-32)
RT_DEV f4 clipVertex(const float* __restrict__ pos, const M4& wvp, float bx, float by) {
  f4 p = mulPoint(mk3(pos[0], pos[1], pos[2]), wvp);
  p.x += bx * p.w;
  p.y += by * p.w;
  return p;
}
// viewport transform + snapping to 8 sub-pixel bits
RT_DEV RVert rasterVertex(f4 p, uint32_t W, uint32_t H) {
  RVert r; r.X = 0; r.Y = 0; r.z = 0.0f; r.ok = false;
  if (!(p.w > 0.0f)) return r;
  const float nx = p.x / p.w, ny = p.y / p.w;
  r.z = p.z / p.w;
  const float sx = (nx + 1.0f) * ((float)W * 0.5f);
  const float sy = (1.0f - ny) * ((float)H * 0.5f);
  const float fx = floorf(sx * 256.0f + 0.5f), fy = floorf(sy * 256.0f + 0.5f);
  if (!(fabsf(fx) < 1073741824.0f) || !(fabsf(fy) < 1073741824.0f)) return r;
  r.X = (long long)fx; r.Y = (long long)fy; r.ok = true;
  return r;
}
// Near-plane clip of one triangle (D3D clips to 0 <= z; the far side is a per-pixel z <= 1 test): 0, 3 or 4 vertices of a
// convex polygon in the input's winding.  New vertices are interpolated in clip space, fp32, from the inside vertex
// towards the outside one, and sit exactly on the plane (z = 0).
RT_DEV int clipNear(const f4 in[3], f4 out[4]) {
  int n = 0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const f4 a = in[k], b = in[(k + 1) % 3];
    const bool ia = a.z >= 0.0f, ib = b.z >= 0.0f;
    if (ia) out[n++] = a;
    if (ia != ib) {
      const f4 p = ia ? a : b, q = ia ? b : a;
      const float t = p.z / (p.z - q.z);
      f4 c; c.x = p.x + (q.x - p.x) * t; c.y = p.y + (q.y - p.y) * t; c.z = 0.0f; c.w = p.w + (q.w - p.w) * t;
      out[n++] = c;
    }
  }
  return n;
}
// Guard band (round 4; the contract is written down in oracle/orc_raster.h): a polygon with a vertex outside |x| <= 256 w, |y| <= 256 w is
// clipped against those four planes in clip space before it is snapped -- the D3D rule behind RayTracer.cpp:751-791; rounds 1-3 dropped
// a triangle whose snapped coordinates left +-2^30.
#define RT_GUARD 256.0f
RT_DEV float guardDistance(const f4& v, int plane) {
  const float gw = RT_GUARD * v.w;
  return plane == 1 ? gw - v.x : plane == 2 ? gw + v.x : plane == 3 ? gw - v.y : gw + v.y;
}
RT_DEV int clipGuard(const f4* in, int n, f4* out, int plane) {
  int m = 0;
  for (int k = 0; k < n; ++k) {
    const f4 a = in[k], b = in[k + 1 == n ? 0 : k + 1];
    const float da = guardDistance(a, plane), db = guardDistance(b, plane);
    const bool ia = da >= 0.0f, ib = db >= 0.0f;
    if (ia) out[m++] = a;
    if (ia != ib) {
      const f4 p = ia ? a : b, q = ia ? b : a;
      const float dp = ia ? da : db, dq = ia ? db : da;
      const float t = dp / (dp - dq);
      f4 c; c.x = p.x + (q.x - p.x) * t; c.y = p.y + (q.y - p.y) * t; c.z = p.z + (q.z - p.z) * t; c.w = p.w + (q.w - p.w) * t;
      const float gw = RT_GUARD * c.w;
      if (plane == 1) c.x = gw; else if (plane == 2) c.x = -gw; else if (plane == 3) c.y = gw; else c.y = -gw;
      out[m++] = c;
    }
  }
  return m;
}
RT_DEV bool outsideGuard(const f4& v) { const float gw = RT_GUARD * v.w; return fabsf(v.x) > gw || fabsf(v.y) > gw; }

RT_DEV bool isTopLeft(long long ax, long long ay, long long bx, long long by) {
  const long long dx = bx - ax, dy = by - ay;
  return (dy == 0 && dx > 0) || dy < 0;
}

// Clears rows of a visibility target and empties the lists the pass appends to.  Since round 3 this is the EXCEPTION: ray generation
// of frame f clears the target of frame f + 2 on its way (raytrace.hip), and this kernel runs only where that has not happened -- a
// context's first two frames, a strip whose rows changed, a caller that rendered visibility without tracing two frames earlier.
__global__ void clearVisDepth(unsigned long long* __restrict__ vd, uint32_t begin, uint32_t end, uint32_t* __restrict__ largeCount, uint32_t* __restrict__ splitCount) {
  if (blockIdx.x == 0 && threadIdx.x == 0) { *largeCount = 0; *splitCount = 0; }      // the large-triangle list of rasterSmall and this set's split list (rayGenKernel) start empty
  // four words (32 bytes) per thread: a quarter of the waves, each with 2 KB of stores in flight
  const uint32_t i = begin + (blockIdx.x * blockDim.x + threadIdx.x) * 4u;
  const unsigned long long clear = RT_VIS_CLEAR;
  if (i + 3u < end && (i & 1u) == 0u) {
    ulonglong2* p = reinterpret_cast<ulonglong2*>(vd + i);
    p[0] = make_ulonglong2(clear, clear); p[1] = make_ulonglong2(clear, clear);
  } else for (uint32_t k = 0; k < 4u; ++k) if (i + k < end) vd[i + k] = clear;
}

// Small triangles (bounding box <= RT_SMALL_BOX pixels), balanced over the lanes of a wave.  Phase 1: one lane per
// triangle does the set-up (transform, snap, cull, box) and parks it in LDS.  Phase 2: the candidate pixels of the
// wave's triangles form one list (prefix sum of the box sizes); lane l takes candidates l, l + 64, ...: finds the
// triangle by binary search in the prefix sums and the pixel inside the box with a multiply-shift division.  The
// per-thread box loop this replaces ran as long as the largest box in the wave (42 us on the 1080p bunny frame).
// Round 4, measured and NOT adopted (profiles/r04_c_pipeline_ab.txt): RT_RASTER_TPW = 16 triangles per wave instead of 64 (four times the
// waves, a quarter of the candidate rounds each) and a fire-and-forget atomicMin per fragment instead of a read of the pixel first
// (RT_RASTER_PREREAD 0).  The kernel is a chain of dependent steps -- indices -> vertices -> set-up -> ~30 candidate pixels per triangle --
// and alone on the chip it falls from 19-21 us to 8.4-8.9; in the frame its four times as many waves take slots from the main stream's
// kernels in one burst instead of trickling beside them, and the FRAME gets 2 % slower (1080p 0.1844 -> 0.188 ms, and the pipeline
// settles into its slower state more often).  The defaults stay at 64 / 1; the variant remains a build option.
#ifndef RT_RASTER_TPW
#define RT_RASTER_TPW 64
#endif
#ifndef RT_RASTER_PREREAD
#define RT_RASTER_PREREAD 1
#endif
#define RT_SMALL_BOX 1024
struct __attribute__((aligned(16))) TriSetup {
  int32_t X[3], Y[3]; float z[3]; uint32_t word;
  int32_t px0, py0; uint32_t bwTl /* box width | top-left flags << 16 */, magic /* ceil(2^24 / box width) */;
  double invA;
};
// The pass's first kernel also carries the frame's constants to the device: `fp` arrives by value (912 bytes of kernel argument, read
// with scalar loads), and block 0 copies it into the constants' device slot `dst` (null: already there) for the kernels that follow.
__global__ void __launch_bounds__(256) rasterSmall(const FrameParams fp, FrameParams* __restrict__ dst, uint32_t rowBegin, uint32_t rowEnd, const float* __restrict__ v0, const uint32_t* __restrict__ i0, uint32_t nt0,
                                                   const float* __restrict__ v1, const uint32_t* __restrict__ i1, uint32_t nt1,
                                                   unsigned long long* __restrict__ vd, LargeTri* __restrict__ large,
                                                   uint32_t* __restrict__ largeCount, uint32_t largeCap, uint32_t* __restrict__ dirty, uint32_t tilesX) {
  constexpr uint32_t TPW = RT_RASTER_TPW;      // triangles per wave: a power of two, 1 .. 64
  __shared__ TriSetup setupMem[4 * TPW];
  __shared__ uint32_t prefixMem[4 * TPW];
  if (blockIdx.x == 0 && dst) {
    const uint32_t* s = reinterpret_cast<const uint32_t*>(&fp);
    uint32_t* d = reinterpret_cast<uint32_t*>(dst);
    for (uint32_t i = threadIdx.x; i < sizeof(FrameParams) / 4; i += blockDim.x) d[i] = s[i];
  }
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  TriSetup* const setup = setupMem + wave * TPW; uint32_t* const prefix = prefixMem + wave * TPW;
  const uint32_t t = (blockIdx.x * 4u + wave) * TPW + lane;
  uint32_t cnt = 0;
  if (lane < TPW && t < nt0 + nt1) {
    const uint32_t inst = t < nt0 ? 0u : 1u;
    const uint32_t prim = inst ? t - nt0 : t;
    const float* verts = inst ? v1 : v0;
    const uint32_t* idx = inst ? i1 : i0;
    // (selected, not indexed: the constants are a kernel argument)
    const M4 wvp0 = cbLoad4x4(fp.po[0].WorldViewProj), wvp1 = cbLoad4x4(fp.po[1].WorldViewProj);
    M4 wvp;
    for (int r = 0; r < 4; ++r) for (int q = 0; q < 4; ++q) wvp.m[r][q] = inst ? wvp1.m[r][q] : wvp0.m[r][q];
    const float bx = inst ? fp.po[1].ProjBias[0] : fp.po[0].ProjBias[0], by = inst ? fp.po[1].ProjBias[1] : fp.po[0].ProjBias[1];
    const uint32_t word = ((inst << 24) | prim) + 1u;
    // One (sub-)triangle in clip space: project, snap, cull, box; small boxes are parked for phase 2 (at most one per
    // lane: sub-triangles of a clipped triangle always go to the tile pass), big ones queued for rasterLarge.
    auto emit = [&](f4 c0, f4 c1, f4 c2, bool toTilePass) {
      const RVert r0 = rasterVertex(c0, fp.W, fp.H), r1 = rasterVertex(c1, fp.W, fp.H), r2 = rasterVertex(c2, fp.W, fp.H);
      if (!(r0.ok && r1.ok && r2.ok)) return;
      const long long X[3] = {r0.X, r1.X, r2.X}, Y[3] = {r0.Y, r1.Y, r2.Y}; const float z[3] = {r0.z, r1.z, r2.z};
      const long long area2 = (X[1] - X[0]) * (Y[2] - Y[0]) - (Y[1] - Y[0]) * (X[2] - X[0]);
      if (area2 <= 0) return;
      const long long minX = min(X[0], min(X[1], X[2])), maxX = max(X[0], max(X[1], X[2]));
      const long long minY = min(Y[0], min(Y[1], Y[2])), maxY = max(Y[0], max(Y[1], Y[2]));
      long long px0 = (minX - 128 + 255) >> 8, px1 = (maxX - 128) >> 8;
      long long py0 = (minY - 128 + 255) >> 8, py1 = (maxY - 128) >> 8;
      px0 = max(px0, 0ll); py0 = max(py0, (long long)rowBegin);
      px1 = min(px1, (long long)fp.W - 1); py1 = min(py1, (long long)rowEnd - 1);
      if (px0 > px1 || py0 > py1) return;
      const long long area = (px1 - px0 + 1) * (py1 - py0 + 1);
      const bool tl0 = isTopLeft(X[1], Y[1], X[2], Y[2]), tl1 = isTopLeft(X[2], Y[2], X[0], Y[0]), tl2 = isTopLeft(X[0], Y[0], X[1], Y[1]);
      const double invA = 1.0 / (double)area2;
      if (toTilePass || area > RT_SMALL_BOX) {
        const uint32_t slot = atomicAdd(largeCount, 1u);
        if (slot < largeCap) {
          LargeTri lt;
          for (int k = 0; k < 3; ++k) { lt.X[k] = (int32_t)X[k]; lt.Y[k] = (int32_t)Y[k]; lt.z[k] = z[k]; }
          lt.word = word; lt.invA = invA; lt.tl = (tl0 ? 1u : 0u) | (tl1 ? 2u : 0u) | (tl2 ? 4u : 0u); lt.pad = 0u;
          large[slot] = lt;
        } else {
          // queue full (more than 65536 big triangles): rasterise it right here -- correct, slow
          const double z0 = (double)z[0], dz1 = (double)z[1] - z0, dz2 = (double)z[2] - z0;
          const int32_t X32[3] = {(int32_t)X[0], (int32_t)X[1], (int32_t)X[2]}, Y32[3] = {(int32_t)Y[0], (int32_t)Y[1], (int32_t)Y[2]};
          for (long long py = py0; py <= py1; ++py)
            for (long long px = px0; px <= px1; ++px) {
              const unsigned long long key = fragmentKey((int32_t)px * 256 + 128, (int32_t)py * 256 + 128, X32, Y32, tl0, tl1, tl2, invA, z0, dz1, dz2, word);
              if (key == ~0ull) continue;
              unsigned long long* dst = vd + (size_t)py * fp.W + (size_t)px;
              if (key < *dst) { atomicMin(dst, key); dirty[(((uint32_t)py - rowBegin) >> 4) * tilesX + ((uint32_t)px >> 4)] = 1u; }
            }
        }
      } else {
        TriSetup ts;
        for (int k = 0; k < 3; ++k) { ts.X[k] = (int32_t)X[k]; ts.Y[k] = (int32_t)Y[k]; ts.z[k] = z[k]; }
        ts.word = word; ts.px0 = (int32_t)px0; ts.py0 = (int32_t)py0;
        const uint32_t bw = (uint32_t)(px1 - px0 + 1);
        ts.bwTl = bw | (tl0 ? 1u << 16 : 0u) | (tl1 ? 1u << 17 : 0u) | (tl2 ? 1u << 18 : 0u);
        ts.magic = ((1u << 24) + bw - 1u) / bw;
        ts.invA = invA;
        setup[lane] = ts;
        cnt = (uint32_t)area;
      }
    };
    f4 cp[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) cp[k] = clipVertex(verts + 6 * (size_t)idx[3 * (size_t)prim + k], wvp, bx, by);
    const bool allIn = cp[0].z >= 0.0f && cp[1].z >= 0.0f && cp[2].z >= 0.0f;
    if (allIn && !(outsideGuard(cp[0]) || outsideGuard(cp[1]) || outsideGuard(cp[2]))) emit(cp[0], cp[1], cp[2], false);
    else {                                            // crosses (or is behind) the near plane, or leaves the guard band: rare
      f4 poly[8], tmp[8];
      int nv = 3;
      if (allIn) { poly[0] = cp[0]; poly[1] = cp[1]; poly[2] = cp[2]; } else nv = clipNear(cp, poly);
      bool guard = false;
      for (int k = 0; k < nv; ++k) guard = guard || outsideGuard(poly[k]);
      if (guard) {
        nv = clipGuard(poly, nv, tmp, 1); nv = clipGuard(tmp, nv, poly, 2);
        nv = clipGuard(poly, nv, tmp, 3); nv = clipGuard(tmp, nv, poly, 4);
      }
      for (int sub = 0; sub + 2 < nv; ++sub) emit(poly[0], poly[sub + 1], poly[sub + 2], true);      // a fan around the first vertex
    }
  }
  // exclusive prefix sum of the candidate counts over the wave
  uint32_t inc = cnt;
  for (int o = 1; o < 64; o <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)inc, o); if ((int)lane >= o) inc += v; }
  if (lane < TPW) prefix[lane] = inc - cnt;
  const uint32_t total = (uint32_t)__shfl((int)inc, 63);
  for (uint32_t w = lane; w < total; w += 64u) {
    uint32_t lo = 0;                       // last triangle whose candidates start at or before w (its count is > 0)
#pragma unroll
    for (uint32_t step = TPW / 2u; step > 0u; step >>= 1) if (prefix[lo + step] <= w) lo += step;
    const TriSetup& S = setup[lo];
    const uint32_t q = w - prefix[lo], bw = S.bwTl & 0xFFFFu;
    const uint32_t ry = (q * S.magic) >> 24, rx = q - ry * bw;       // q / bw exactly: q < 1024, bw <= 1024
    const int32_t px = S.px0 + (int32_t)rx, py = S.py0 + (int32_t)ry;
    const int32_t X[3] = {S.X[0], S.X[1], S.X[2]}, Y[3] = {S.Y[0], S.Y[1], S.Y[2]};
    const double z0 = (double)S.z[0], dz1 = (double)S.z[1] - z0, dz2 = (double)S.z[2] - z0;
    const unsigned long long key = fragmentKey(px * 256 + 128, py * 256 + 128, X, Y, (S.bwTl >> 16) & 1u, (S.bwTl >> 17) & 1u, (S.bwTl >> 18) & 1u,
                                               S.invA, z0, dz1, dz2, S.word);
    if (key == ~0ull) continue;
    unsigned long long* dst = vd + (size_t)py * fp.W + (size_t)px;
#if RT_RASTER_PREREAD
    // (the tile's word, rtggx_context.h visDirtyBuf: set by whoever finds the pixel still clear -- the first atomic on a pixel comes from such a lane)
    const unsigned long long before = *dst;
    if (key < before) { atomicMin(dst, key); if (before == RT_VIS_CLEAR) dirty[(((uint32_t)py - rowBegin) >> 4) * tilesX + ((uint32_t)px >> 4)] = 1u; }
#else
    atomicMin(dst, key);      // (no result used: the compiler emits the no-return form, nothing waits for it)
    dirty[(((uint32_t)py - rowBegin) >> 4) * tilesX + ((uint32_t)px >> 4)] = 1u;
#endif
  }
}

// The queued large triangles merged into the target: 64 x 16 pixels per workgroup, four per lane (rows y, y + 4, y + 8, y + 12).  One
// pixel per lane made this a kernel of 32 400 waves at 1080p that each fetched the list, rejected most of it by bounding box and left:
// 30 M wave quad-cycles per frame for 8 us of work (profiles/r03_k_pmc_report.txt); a quarter of the waves do the same work.
#ifndef RT_LARGE_ROWS
#define RT_LARGE_ROWS 16
#endif
__global__ void __launch_bounds__(256) rasterLarge(uint32_t W, uint32_t rowBegin, uint32_t rowEnd, unsigned long long* __restrict__ vd, const LargeTri* __restrict__ large,
                                                   const uint32_t* __restrict__ largeCount, uint32_t largeCap, uint32_t* __restrict__ dirty, uint32_t tilesX) {
  const uint32_t px = blockIdx.x * 64 + (threadIdx.x & 63);
  const uint32_t y0 = rowBegin + blockIdx.y * RT_LARGE_ROWS, py0 = y0 + (threadIdx.x >> 6);
  const uint32_t n = min(*largeCount, largeCap);
  const int32_t bx0 = (int32_t)(blockIdx.x * 64) * 256 + 128, bx1 = bx0 + 63 * 256, by0 = (int32_t)y0 * 256 + 128, by1 = by0 + (RT_LARGE_ROWS - 1) * 256;
  const int32_t PX = (int32_t)px * 256 + 128;
  unsigned long long best[RT_LARGE_ROWS / 4];
#pragma unroll
  for (int k = 0; k < RT_LARGE_ROWS / 4; ++k) best[k] = ~0ull;
  for (uint32_t i = 0; i < n; ++i) {
    const LargeTri lt = large[i];
    const int32_t minX = min(lt.X[0], min(lt.X[1], lt.X[2])), maxX = max(lt.X[0], max(lt.X[1], lt.X[2]));
    const int32_t minY = min(lt.Y[0], min(lt.Y[1], lt.Y[2])), maxY = max(lt.Y[0], max(lt.Y[1], lt.Y[2]));
    if (maxX < bx0 || minX > bx1 || maxY < by0 || minY > by1) continue;   // uniform per workgroup
    const double z0 = (double)lt.z[0], dz1 = (double)lt.z[1] - z0, dz2 = (double)lt.z[2] - z0;
#pragma unroll
    for (int k = 0; k < RT_LARGE_ROWS / 4; ++k) {
      const unsigned long long key = fragmentKey(PX, (int32_t)(py0 + 4u * k) * 256 + 128, lt.X, lt.Y, lt.tl & 1u, (lt.tl >> 1) & 1u, (lt.tl >> 2) & 1u, lt.invA, z0, dz1, dz2, lt.word);
      best[k] = key < best[k] ? key : best[k];
    }
  }
  if (px >= W) return;
#pragma unroll
  for (int k = 0; k < RT_LARGE_ROWS / 4; ++k) {
    const uint32_t py = py0 + 4u * k;
    if (py < rowEnd && best[k] != ~0ull) {
      unsigned long long* dst = vd + (size_t)py * W + px;
      if (best[k] < *dst) { *dst = best[k]; dirty[((py - rowBegin) >> 4) * tilesX + (px >> 4)] = 1u; }
    }
  }
}

__global__ void unpackVisDepthKernel(const unsigned long long* __restrict__ vd, uint32_t* __restrict__ vis, uint32_t* __restrict__ depth, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const unsigned long long k = vd[i]; vis[i] = (uint32_t)k; depth[i] = (uint32_t)(k >> 32); }
}
__global__ void packVisDepthKernel(unsigned long long* __restrict__ vd, const uint32_t* __restrict__ vis, const uint32_t* __restrict__ depth, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) vd[i] = ((unsigned long long)depth[i] << 32) | vis[i];
}

int launchVisibility(rtggx_context* c, const FrameParams& fp, hipStream_t s, hipEvent_t done) {
  uint32_t rb, re;
  passRows(fp, ROWS_GBUFFER, rb, re);
  const uint32_t begin = rb * fp.W, end = re * fp.W;
  if (end <= begin) return 0;
  // the target was cleared for this frame by the ray generation two frames back -- unless it was not (see clearVisDepth)
  const uint32_t target = c->frameCounter % RT_VIS_RING;
  auto& vc = c->visClearedAt[target];
  auto& vf = c->visFlags[target];
  uint32_t* const dirty = c->visDirtyBuf[target];
  const uint32_t tilesX = (fp.W + 15) / 16, tilesY = (re - rb + 15) / 16;
  if (!(vc.frame == c->frameCounter && vc.rows[0] <= rb && vc.rows[1] >= re)) {
    hipLaunchKernelGGL(clearVisDepth, dim3((end - begin + 1023) / 1024), dim3(256), 0, s, c->visDepth, begin, end, c->largeCount, c->splitCount);
    RT_HIP(hipMemsetAsync(dirty, 0, (size_t)tilesX * tilesY * 4, s));
    ++c->visStandaloneClears;
  }
  // (cleared ahead over a superset of these rows: every word the clear knew of is 0 and every pixel clear, whatever the tiles' origin)
  vf.rows[0] = rb; vf.rows[1] = re; vf.rasterFrame = c->frameCounter;
  vc.frame = 0u;
  const uint32_t nt = c->mesh[0].numTris + c->mesh[1].numTris;
  FrameParams* const dst = c->slotUploaded ? (FrameParams*)nullptr : c->dParams + c->slot;
  c->slotUploaded = true;
  // (with no triangles at all the kernel still runs, for the constants)
  const dim3 grid(nt ? (nt + 4u * RT_RASTER_TPW - 1u) / (4u * RT_RASTER_TPW) : 1u);
  hipLaunchKernelGGL(rasterSmall, grid, dim3(256), 0, s, fp, dst, rb, re, (const float*)c->mesh[0].verts, (const uint32_t*)c->mesh[0].indices, c->mesh[0].numTris,
                     (const float*)c->mesh[1].verts, (const uint32_t*)c->mesh[1].indices, c->mesh[1].numTris, c->visDepth, (LargeTri*)c->largeTris, c->largeCount, c->largeCapacity, dirty, tilesX);
  if (nt) {
    const dim3 lgrid((fp.W + 63) / 64, (re - rb + RT_LARGE_ROWS - 1) / RT_LARGE_ROWS);
    if (done && c->attachEvents) {      // the event rides on the pass's last kernel (rtggx_context.h)
      hipExtLaunchKernelGGL(rasterLarge, lgrid, dim3(256), 0, s, nullptr, done, 0, fp.W, rb, re, c->visDepth, (const LargeTri*)c->largeTris, (const uint32_t*)c->largeCount, c->largeCapacity, dirty, tilesX);
      done = nullptr;
    } else hipLaunchKernelGGL(rasterLarge, lgrid, dim3(256), 0, s, fp.W, rb, re, c->visDepth, (const LargeTri*)c->largeTris, (const uint32_t*)c->largeCount, c->largeCapacity, dirty, tilesX);
  }
  if (done) hipEventRecord(done, s);
  RT_HIP(hipGetLastError());
  return 0;
}

int unpackVisDepth(rtggx_context* c, uint32_t* dVis, uint32_t* dDepth, hipStream_t s) {
  const uint32_t n = c->W * c->H;
  hipLaunchKernelGGL(unpackVisDepthKernel, dim3((n + 255) / 256), dim3(256), 0, s, c->visDepth, dVis, dDepth, n);
  RT_HIP(hipGetLastError());
  return 0;
}
int packVisDepth(rtggx_context* c, const uint32_t* dVis, const uint32_t* dDepth, hipStream_t s) {
  const uint32_t n = c->W * c->H;
  // a caller's visibility: every tile may hold something
  RT_HIP(hipMemsetAsync(c->visDirtyBuf[c->frameCounter % RT_VIS_RING], 0xFF, (size_t)((c->W + 15) / 16) * ((c->H + 15) / 16 + 1) * 4, s));
  hipLaunchKernelGGL(packVisDepthKernel, dim3((n + 255) / 256), dim3(256), 0, s, c->visDepth, dVis, dDepth, n);
  RT_HIP(hipGetLastError());
  return 0;
}

}  // namespace rt
