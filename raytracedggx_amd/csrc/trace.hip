// Kernel 2 of the ray-tracing pass: TraceRay (RayTracedGGX/Content/Shaders/RayTracing.hlsl:183-198) over the ray
// bins filled by rayGenKernel (raytrace.hip, rt_queue.h).
//
// One lane = one ray, one wave = one bin: the (usually <= 64) rays of one 8x8 pixel sub-tile, which
// start next to each other and mostly walk the same nodes.  Every traversal step is a DEPENDENT random 64-byte
// fetch (a BVH node with both child boxes, or a leaf triangle), so the kernel is bound by the latency of those
// fetches, not by arithmetic or bandwidth (tools/microbench/gather.hip, profiles/r01_*).  What the measurements
// settled:
//   * exactly ONE round trip to memory per step: the record of a lane -- node or triangle -- is fetched in one
//     phase and pinned in registers before the type branch (hipcc otherwise narrows the loads per use and sinks
//     them behind the branch: three round trips per step);
//   * no persistence, no work queue: the rays of this workload are short (~16 node visits and ~1.5 triangle tests
//     each) and there are only ~2 of them per resident lane, so a persistent kernel that re-deals rays to idle
//     lanes spent more on dealing (atomics on list heads, refill fetches, tail imbalance) than it saved: 0.55 ms
//     against 0.22 ms for this kernel on the 1080p bunny frame (profiles/r01_c).  The hardware dispatcher balances
//     the ~10^4 short-lived waves over the CUs;
//   * the world-space ray stays in registers as two object-space rays: switching from the ground instance to the
//     model costs no fetch.
// Per-lane traversal stack in LDS ([entry][lane] layout: conflict-free ds_read/ds_write_b32), RT_STACK entries,
// deeper pushes spill to global memory (launchTrace sizes the spill area from the depth of the built trees; the
// bunny and dragon trees never need it).  Semantics (DESIGN.md "Traversal"): two-level, rays carried into each
// instance's object space, nearer child first, watertight ray/triangle test (Woop, Benthin, Wald 2013), no
// culling, TMin < t < TMax, ties to the lower (instance, primitive).
#include "rt_queue.h"

namespace rt {

#define RT_STACK 16          // LDS stack entries per lane (deepest stack seen on the bunny/dragon frames: 12)

struct TraceArgs {
  const float4* nodes0; const float4* tris0;   // 64-byte records: 4 x float4 each
  const float4* nodes1; const float4* tris1;
  int32_t root0, root1;
  uint32_t haveMesh0, haveMesh1;
  const RayRec* rays; HitRec* hits;
  const uint32_t* binCount; uint32_t numBins;
  int32_t* overflow;        // [entry][numBinsMax * RT_BIN] spill area for stacks deeper than RT_STACK
  uint32_t* rayTotals;       // 256 per-frame partial counters (+ RT_TRACE_STATS words from 256)
  uint32_t countRowBegin, countRowEnd, width;
  size_t spillStride;
};

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

// Wave w of the grid traces the rays of bin w, 64 at a time (a second round only where a sub-tile has more than 64
// rays, i.e. diffuse rays besides the reflection rays).
__global__ void __launch_bounds__(256) traceKernel(const FrameParams* __restrict__ fpp, TraceArgs A) {
  __shared__ int32_t stackMem[RT_STACK * 256];
  const FrameParams& fp = *fpp;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t bin = blockIdx.x * 4u + wave;
  if (bin >= A.numBins) return;
  const uint32_t count = min(A.binCount[bin], RT_BIN);
  if (count == 0u) return;
  int32_t* const stack = stackMem + wave * (RT_STACK * 64) + lane;                      // entry e at stack[e * 64]
  const size_t spillStride = A.spillStride;
  int32_t* const spill = A.overflow + bin * RT_BIN + lane;                               // entry e at spill[e * spillStride]
  uint32_t nRays = 0;
#ifdef RT_TRACE_STATS
  uint32_t stNode = 0, stLeaf = 0, stIter = 0, stDeep = 0;
#endif
  for (uint32_t base = 0; base < count; base += 64u) {
  const uint32_t slot = bin * RT_BIN + base + lane;
  bool active = base + lane < count;

  // ---- the ray: world space -> the object spaces of both instances --------------------------------------------
  const float4* rp = reinterpret_cast<const float4*>(A.rays + (active ? slot : bin * RT_BIN));
  const float4 ra = rp[0], rb = rp[1];
  const uint4 rc = reinterpret_cast<const uint4*>(rp)[2];
  const float tmin = ra.w;
  float bestT = rb.w, bestB1 = 0.0f, bestB2 = 0.0f;
  uint32_t bestId = 0xFFFFFFFFu;
  const uint32_t skip = rc.y;
  uint32_t inst = A.haveMesh0 ? 0u : 1u;
  const LaneRay r1 = toObject(ra.x, ra.y, ra.z, rb.x, rb.y, rb.z, fp.invWorld[1]);
  LaneRay r = inst ? r1 : toObject(ra.x, ra.y, ra.z, rb.x, rb.y, rb.z, fp.invWorld[0]);
  int32_t cur = inst ? A.root1 : A.root0;
  int sp = 0;
  {
    const uint32_t row = rc.x / A.width;
    if (active && row >= A.countRowBegin && row < A.countRowEnd) ++nRays;
  }
  if (active && (!(bestT > tmin) || (inst == 1u && A.haveMesh1 == 0u))) {   // degenerate interval / empty scene: a miss
    HitRec h; h.t = bestT; h.b1 = 0.0f; h.b2 = 0.0f; h.id = 0xFFFFFFFFu; A.hits[slot] = h;
    active = false;
  }

  // ---- traversal -----------------------------------------------------------------------------------------------
  while (__ballot(active)) {
#ifdef RT_TRACE_STATS
    ++stIter; if (active) { if (cur < 0) ++stLeaf; else ++stNode; }
#endif
    if (active) {
      bool needPop = false;
      // ONE fetch phase per step: the 64-byte record of this lane, node or leaf triangle alike
      const bool leaf = cur < 0;
      const float4* rec = leaf ? (inst ? A.tris1 : A.tris0) + (size_t)(~cur) * 4 : (inst ? A.nodes1 : A.nodes0) + (size_t)cur * 4;
      float4 n0 = rec[0], n1 = rec[1], n2 = rec[2], n3 = rec[3];
      asm volatile("" : "+v"(n0.x), "+v"(n0.y), "+v"(n0.z), "+v"(n0.w), "+v"(n1.x), "+v"(n1.y), "+v"(n1.z), "+v"(n1.w));
      asm volatile("" : "+v"(n2.x), "+v"(n2.y), "+v"(n2.z), "+v"(n2.w), "+v"(n3.x), "+v"(n3.y));
      const int32_t w12 = __float_as_int(n3.x);             // node: left child; triangle: primitive id
      if (!leaf) {
        // n0 = lmin.xyz lmax.x | n1 = lmax.yz rmin.xy | n2 = rmin.z rmax.xyz | n3 = left right pad pad
        float ln, lf, rn, rf;
        slabTest(r, n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, tmin, bestT, ln, lf);
        slabTest(r, n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, tmin, bestT, rn, rf);
        const bool hl = ln <= lf * 1.0000004f, hr = rn <= rf * 1.0000004f;
        const int32_t left = w12, right = __float_as_int(n3.y);
        if (hl && hr) {
          const bool leftFirst = ln <= rn;
          const int32_t far = leftFirst ? right : left;
          if (sp < RT_STACK) stack[sp * 64] = far; else spill[(size_t)(sp - RT_STACK) * spillStride] = far;
          ++sp;
#ifdef RT_TRACE_STATS
          if ((uint32_t)sp > stDeep) stDeep = (uint32_t)sp;
#endif
          cur = leftFirst ? left : right;
        } else if (hl) cur = left;
        else if (hr) cur = right;
        else needPop = true;
      } else {
        // n0 = v0.xyz v1.x | n1 = v1.yz v2.xy | n2 = v2.z pad pad pad | n3 = prim pad pad pad
        const uint32_t id = (inst << 24) | (uint32_t)w12;
        if (id != skip) {
          float t, b1, b2;
          if (woopTest(r, n0, n1, n2, t, b1, b2) && t > tmin) {
            const bool closer = t < bestT;
            const bool tie = bestId != 0xFFFFFFFFu && t == bestT && id < bestId;
            if (closer || tie) { bestT = t; bestId = id; bestB1 = b1; bestB2 = b2; }
          }
        }
        needPop = true;
      }
      if (needPop) {
        if (sp > 0) {
          --sp;
          if (sp < RT_STACK) cur = stack[sp * 64]; else cur = spill[(size_t)(sp - RT_STACK) * spillStride];
        } else if (inst == 0u && A.haveMesh1 != 0u) {
          inst = 1u; r = r1; cur = A.root1;                  // ground done: continue in the model's object space
        } else {
          HitRec h; h.t = bestT; h.b1 = bestB1; h.b2 = bestB2; h.id = bestId;
          A.hits[slot] = h;
          active = false;
        }
      }
    }
  }
  }   // base

  // ray statistics: one fire-and-forget atomic per wave, spread over 256 words
  for (int o = 32; o > 0; o >>= 1) nRays += __shfl_down(nRays, o);
  if (lane == 0 && nRays) atomicAdd(&A.rayTotals[bin & 255u], nRays);
#ifdef RT_TRACE_STATS
  for (int o = 32; o > 0; o >>= 1) { stNode += __shfl_down(stNode, o); stLeaf += __shfl_down(stLeaf, o); stDeep = max(stDeep, (uint32_t)__shfl_down((int)stDeep, o)); }
  if (lane == 0) {   // lane node steps, lane leaf steps, wave iterations, waves, deepest stack
    atomicAdd(&A.rayTotals[256], stNode); atomicAdd(&A.rayTotals[257], stLeaf); atomicAdd(&A.rayTotals[258], stIter);
    atomicAdd(&A.rayTotals[259], 1u); atomicMax(&A.rayTotals[260], stDeep);
  }
#endif
}

int launchTrace(rtggx_context* c, const FrameParams& fp, hipStream_t s, uint32_t numBins, bool countRays) {
  TraceArgs T;
  if (numBins == 0) return 0;
  if (numBins > c->numBinsMax) { setError("launchTrace: %u bins exceed the %u allocated", numBins, c->numBinsMax); return -1; }
  const bool have0 = c->mesh[0].tris != nullptr, have1 = c->mesh[1].tris != nullptr;
  // a mesh with one triangle has no internal nodes; an absent mesh has nothing: point those bases at the dummy record
  T.nodes0 = (const float4*)(have0 && c->mesh[0].nodes ? (const void*)c->mesh[0].nodes : c->dummyRecord);
  T.tris0 = (const float4*)(have0 ? (const void*)c->mesh[0].tris : c->dummyRecord);
  T.nodes1 = (const float4*)(have1 && c->mesh[1].nodes ? (const void*)c->mesh[1].nodes : c->dummyRecord);
  T.tris1 = (const float4*)(have1 ? (const void*)c->mesh[1].tris : c->dummyRecord);
  T.root0 = c->mesh[0].root; T.root1 = c->mesh[1].root; T.haveMesh0 = have0; T.haveMesh1 = have1;
  T.rays = (const RayRec*)c->rayQueue; T.hits = (HitRec*)c->hitQueue;
  T.binCount = c->binCount; T.numBins = numBins;
  // stacks deeper than the LDS part spill to global memory; the built trees say how deep they can get
  const uint32_t deepest = c->mesh[0].depth > c->mesh[1].depth ? c->mesh[0].depth : c->mesh[1].depth;
  if (deepest > RT_STACK + c->spillEntries) {
    RT_HIP(hipStreamSynchronize(s));
    if (c->stackOverflow) { RT_HIP(hipFree(c->stackOverflow)); c->stackOverflow = nullptr; }
    c->spillEntries = deepest - RT_STACK;
    RT_HIP(hipMalloc(&c->stackOverflow, (size_t)c->spillEntries * c->numBinsMax * RT_BIN * 4));
  }
  T.overflow = c->stackOverflow; T.rayTotals = c->rayCounter32;
  T.spillStride = (size_t)c->numBinsMax * RT_BIN;
  T.countRowBegin = countRays ? fp.rowBegin : 0u; T.countRowEnd = countRays ? fp.rowEnd : 0u; T.width = fp.W;
  hipLaunchKernelGGL(traceKernel, dim3((numBins + 3u) / 4u), dim3(256), 0, s, c->dParams + c->slot, T);
  RT_HIP(hipGetLastError());
  return 0;
}

}  // namespace rt
