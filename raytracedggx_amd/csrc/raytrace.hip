// Ray-tracing pass: the gfx950 replacement of DispatchRays(W,H,1) over RayTracing.cso
// (RayTracer::rayTrace, RayTracedGGX/Content/RayTracer.cpp:793-810).  The four DXR shaders of
// RayTracedGGX/Content/Shaders/RayTracing.hlsl become three kernels around a ray queue:
//
//   rayGenKernel   raygenMain :541-565 up to the TraceRay calls: G-buffer reconstruction from the
//                  visibility buffer (getPrimarySurface :277-333), GGX / uniform-sphere sampling
//                  (:92-162, :394-406), BRDF weight (:459-480).  Pixels whose ray is degenerate
//                  (background, NoL <= 0) are finished here (missMain :620-625 / zero); every real
//                  ray is compacted, by wave ballot, into the bin of the wave's own 8x8-pixel sub-tile
//                  (rt_queue.h: no atomics).
//   traceKernel    TraceRay :183-198 (trace.hip): one wave per bin over a two-level software BVH -- the ray
//                  is carried into each instance's object space (TLAS = two world->object matrices) and
//                  walks the 4-wide collapse of the Morton-ordered PLOC tree with a per-lane stack in LDS,
//                  nearer child first, lanes sharing work inside the wave.  Ray/triangle: watertight test
//                  (Woop, Benthin, Wald 2013), no culling, TMin < t < TMax, ties to the lower
//                  (instance, primitive).
//   shadeKernel    closestHitReflection :571-590, closestHitDiffuse :593-614, missMain :620-625 and
//                  the tail of computeReflection / computeDiffuse, with Material.hlsli,
//                  BRDFModels.hlsli, SHIrradianceTypeless.hlsli:16-37; writes RayTracingOut0/1.
//
// rayGenKernel and traceKernel run on stream B, shadeKernel on the main stream one frame behind (capi.hip).
// Roofline: HBM by decree of the metric (no MFMA work exists here).  Algorithmic bytes:
// rayGen 18 B/pixel (+64 B per queued ray); trace 64 B ray + 16 B hit per ray + the scene arrays
// once; shade 64+16 B in, 4 B out per ray.  The BVH (<= 14 MB) is L2/MALL resident, so traversal is
// bound by L1 request rate, issue and latency, not by HBM (DESIGN.md "Roofline").
#include <hip/hip_ext.h>
#include "rt_queue.h"
#include "rt_traverse.h"

namespace rt {

#define RT_PI 3.1415926535897f   // BRDFModels.hlsli:5

// ---- environment (RayTracing.hlsl:167-180; D3D cube sampling restated) --------------
struct EnvRef { const uint2* __restrict__ texels; uint32_t size, mips; const uint32_t* __restrict__ mipOffset; };

RT_DEV void cubeFaceUV(f3 d, int& face, float& u, float& v) {
  const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
  if (ax >= ay && ax >= az) { face = d.x >= 0.0f ? 0 : 1; u = (d.x >= 0.0f ? -d.z : d.z) / ax; v = -d.y / ax; }
  else if (ay >= az) { face = d.y >= 0.0f ? 2 : 3; u = d.x / ay; v = (d.y >= 0.0f ? d.z : -d.z) / ay; }
  else { face = d.z >= 0.0f ? 4 : 5; u = (d.z >= 0.0f ? d.x : -d.x) / az; v = -d.y / az; }
}
RT_DEV f3 cubeFaceDir(int face, float u, float v) {
  switch (face) {
    case 0: return mk3(1.0f, -v, -u);
    case 1: return mk3(-1.0f, -v, u);
    case 2: return mk3(u, 1.0f, v);
    case 3: return mk3(u, -1.0f, -v);
    case 4: return mk3(u, -v, 1.0f);
    default: return mk3(-u, -v, -1.0f);
  }
}
// MIP0: the level is known to be 0 where it is called (the sky behind a pixel, a ray that misses): its texels start at offset 0, and
// the offset table -- one more dependent fetch in front of every texel -- is not read.
template <bool MIP0 = false>
RT_DEV f3 cubeTexel(const EnvRef& e, uint32_t mip, int face, int x, int y) {
  const int s = (int)((e.size >> mip) ? (e.size >> mip) : 1u);
  if (x < 0 || y < 0 || x >= s || y >= s) {
    const float u = ((float)x + 0.5f) / (float)s * 2.0f - 1.0f, v = ((float)y + 0.5f) / (float)s * 2.0f - 1.0f;
    float uu, vv; cubeFaceUV(cubeFaceDir(face, u, v), face, uu, vv);
    x = (int)floorf((uu * 0.5f + 0.5f) * (float)s); y = (int)floorf((vv * 0.5f + 0.5f) * (float)s);
    x = min(max(x, 0), s - 1); y = min(max(y, 0), s - 1);
  }
  const uint2 t = e.texels[(MIP0 ? 0u : e.mipOffset[mip]) + (uint32_t)face * (uint32_t)(s * s) + (uint32_t)(y * s + x)];
  return mk3(f16ToF32(t.x & 0xFFFFu), f16ToF32(t.x >> 16), f16ToF32(t.y & 0xFFFFu));
}
template <bool MIP0 = false>
RT_DEV f3 cubeBilinear(const EnvRef& e, uint32_t mip, int face, float u, float v) {
  const int s = (int)((e.size >> mip) ? (e.size >> mip) : 1u);
  const float x = (u * 0.5f + 0.5f) * (float)s - 0.5f, y = (v * 0.5f + 0.5f) * (float)s - 0.5f;
  const float x0 = floorf(x), y0 = floorf(y);
  const float fx = x - x0, fy = y - y0;
  const int ix = (int)x0, iy = (int)y0;
  const float w00 = (1.0f - fx) * (1.0f - fy), w10 = fx * (1.0f - fy), w01 = (1.0f - fx) * fy, w11 = fx * fy;
  const f3 c00 = cubeTexel<MIP0>(e, mip, face, ix, iy), c10 = cubeTexel<MIP0>(e, mip, face, ix + 1, iy);
  const f3 c01 = cubeTexel<MIP0>(e, mip, face, ix, iy + 1), c11 = cubeTexel<MIP0>(e, mip, face, ix + 1, iy + 1);
  return ((c00 * w00 + c10 * w10) + c01 * w01) + c11 * w11;
}
// environment(e, dir, 0): RayTracing.hlsl:620-625 (missMain) and the sky behind a pixel -- the same arithmetic with the level's constants folded
RT_DEV f3 environmentLevel0(const EnvRef& e, f3 dir) {
  int face; float u, v; cubeFaceUV(dir, face, u, v);
  return cubeBilinear<true>(e, 0u, face, u, v);
}
__device__ __noinline__ f3 environment(EnvRef e, f3 dir, float level) {
  int face; float u, v; cubeFaceUV(dir, face, u, v);
  const float maxLevel = (float)(e.mips - 1);
  const float l = fminf(fmaxf(level, 0.0f), maxLevel);
  const float l0 = floorf(l), fl = l - l0;
  const uint32_t m0 = (uint32_t)l0, m1 = min(m0 + 1, e.mips - 1);
  const f3 a = cubeBilinear(e, m0, face, u, v);
  if (fl == 0.0f) return a;
  const f3 b = cubeBilinear(e, m1, face, u, v);
  return a * (1.0f - fl) + b * fl;
}

// ---- SHIrradianceTypeless.hlsli:16-37 ------------------------------------------------------------------
RT_DEV f3 evaluateSHIrradiance(const float* __restrict__ sh, f3 norm) {
  const float c1 = 0.42904276540489171563379376569857f, c2 = 0.51166335397324424423977581244463f;
  const float c3 = 0.24770795610037568833406429782001f, c4 = 0.88622692545275801364908374167057f;
  const float x = -norm.x, y = -norm.y, z = norm.z;
#define SHL(i) mk3(sh[3 * (i)], sh[3 * (i) + 1], sh[3 * (i) + 2])
  f3 irr = (c1 * (x * x - y * y)) * SHL(8);
  irr = irr + (c3 * (3.0f * z * z - 1.0f)) * SHL(6);
  irr = irr + c4 * SHL(0);
  irr = irr + (2.0f * c1) * ((SHL(4) * x * y + SHL(7) * x * z) + SHL(5) * y * z);
  irr = irr + (2.0f * c2) * ((SHL(3) * x + SHL(1) * y) + SHL(2) * z);
#undef SHL
  return mk3(fmaxf(0.0f, irr.x), fmaxf(0.0f, irr.y), fmaxf(0.0f, irr.z));
}

// ---- Material.hlsli -----------------------------------------------------------------------------------
RT_DEV f2 getUV(f3 n, f3 p, f3 scl) {   // :16-23
  f2 uv; uv.x = fabsf(n.x) * p.y * scl.y; uv.y = fabsf(n.x) * p.z * scl.z;
  uv.x += fabsf(n.y) * p.z * scl.z; uv.y += fabsf(n.y) * p.x * scl.x;
  uv.x += fabsf(n.z) * p.x * scl.x; uv.y += fabsf(n.z) * p.y * scl.y;
  uv.x = uv.x * 0.5f + 0.5f; uv.y = uv.y * 0.5f + 0.5f;
  return uv;
}
RT_DEV f2 getRoughMetal(const RtggxCBMaterial& mat, uint32_t inst, f2 uv) {   // :30-48
  float rough = mat.RoughMetals[inst][0];
  if (inst == 0) {
    const uint32_t px = ftou(uv.x * 5.0f) & 1u, py = ftou(uv.y * 5.0f) & 1u;
    rough = (px ^ py) ? rough * 0.25f : rough;
  }
  f2 r; r.x = rough; r.y = mat.RoughMetals[inst][1];
  return r;
}

// ---- BRDFModels.hlsli ---------------------------------------------------------------------------------
RT_DEV float visSmith(float roughness, float NoV, float NoL) {   // :30-39
  const float a = roughness * roughness, a2 = a * a;
  const float v = NoV + sqrtf(NoV * (NoV - NoV * a2) + a2);
  const float l = NoL + sqrtf(NoL * (NoL - NoL * a2) + a2);
  return 1.0f / (v * l);
}
RT_DEV f3 fSchlick(f3 spec, float VoH) {   // :54-62, pow(x,5) by multiplication
  const float x = 1.0f - VoH, x2 = x * x;
  const float fc = (x2 * x2) * x;
  const float s = saturatef(50.0f * spec.y) * fc;
  return mk3(s + (1.0f - fc) * spec.x, s + (1.0f - fc) * spec.y, s + (1.0f - fc) * spec.z);
}
RT_DEV f3 envBRDFApprox(f3 spec, float roughness, float NoV) {   // :64-77
  const float rx = roughness * -1.0f + 1.0f, ry = roughness * -0.0275f + 0.0425f;
  const float rz = roughness * -0.572f + 1.04f, rw = roughness * 0.022f + -0.04f;
  const float a004 = fminf(rx * rx, exp2Contract(-9.28f * NoV)) * rx + ry;
  const float ABx = -1.04f * a004 + rz;
  float ABy = 1.04f * a004 + rw;
  ABy *= saturatef(50.0f * spec.y);
  return mk3(spec.x * ABx + ABy, spec.y * ABx + ABy, spec.z * ABx + ABy);
}

// ---- RayTracing.hlsl helpers ---------------------------------------------------------------------------
RT_DEV uint32_t rng(uint32_t seed) {   // :379-387
  seed = seed * 747796405u + 1u;
  seed = ((seed >> ((seed >> 28) + 4u)) ^ seed) * 277803737u;
  seed = (seed >> 22) ^ seed;
  return seed;
}
RT_DEV float calcMipFromRoughness(float rgh, float mipCount) {   // :416-422
  const float level = 3.0f - 1.15f * log2Contract(rgh);
  return mipCount - 1.0f - level;
}
RT_DEV f3 localToWorld(f3 n, f3 l) {   // :129-147
  const f3 up = fabsf(n.y) < 0.999f ? mk3(0.0f, 1.0f, 0.0f) : mk3(1.0f, 0.0f, 0.0f);
  const f3 xAxis = normalize3(cross3(up, n));
  const f3 yAxis = cross3(n, xAxis);
  return (xAxis * l.x + yAxis * l.y) + n * l.z;
}
// Opt-in sampler (rtggx_set_sampler; north_star: "GGX-VNDF importance sampling in the hit path"): the half vector from the distribution of
// VISIBLE normals (Heitz 2018, "Sampling the GGX Distribution of Visible Normals", listing 1; isotropic, alpha = roughness^2 as in
// computeLocalDirectionGGX), in the tangent frame of localToWorld.  The reference samples the plain GGX NDF (RayTracing.hlsl:92-101,
// 424-484): that stays the default and the parity path; this one has its own oracle counterpart (orc_raytrace.h vndf_half_vector).
RT_DEV f3 vndfHalfVector(f3 n, f3 v, float alpha, float cosPhi, float sinPhi, float u) {
  const f3 up = fabsf(n.y) < 0.999f ? mk3(0.0f, 1.0f, 0.0f) : mk3(1.0f, 0.0f, 0.0f);
  const f3 xAxis = normalize3(cross3(up, n));
  const f3 yAxis = cross3(n, xAxis);
  const f3 ve = mk3(dot3(v, xAxis), dot3(v, yAxis), dot3(v, n));
  const f3 vh = normalize3(mk3(alpha * ve.x, alpha * ve.y, ve.z));
  const float lensq = vh.x * vh.x + vh.y * vh.y;
  const f3 t1v = lensq > 0.0f ? mk3(-vh.y, vh.x, 0.0f) * (1.0f / sqrtf(lensq)) : mk3(1.0f, 0.0f, 0.0f);
  const f3 t2v = cross3(vh, t1v);
  const float r = sqrtf(u);
  const float t1 = r * cosPhi;
  float t2 = r * sinPhi;
  const float sw = 0.5f * (1.0f + vh.z);
  t2 = (1.0f - sw) * sqrtf(1.0f - t1 * t1) + sw * t2;
  const f3 nh = (t1 * t1v + t2 * t2v) + sqrtf(fmaxf(0.0f, (1.0f - t1 * t1) - t2 * t2)) * vh;
  const f3 hl = normalize3(mk3(alpha * nh.x, alpha * nh.y, fmaxf(0.0f, nh.z)));
  return (xAxis * hl.x + yAxis * hl.y) + n * hl.z;
}
struct Tri3 { f3 pos[3], nrm[3]; };
RT_DEV Tri3 getVertices(const float4* __restrict__ fat, uint32_t prim) {   // :230-244, from the primitive's fat triangle (rtggx_context.h): five 16-byte loads, one dependent step
  const float4* p = fat + 5 * (size_t)prim;
  const float4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
  Tri3 v;
  v.pos[0] = mk3(a.x, a.y, a.z); v.nrm[0] = mk3(a.w, b.x, b.y);
  v.pos[1] = mk3(b.z, b.w, c.x); v.nrm[1] = mk3(c.y, c.z, c.w);
  v.pos[2] = mk3(d.x, d.y, d.z); v.nrm[2] = mk3(d.w, e.x, e.y);
  return v;
}
struct Attrib { f3 Pos, Nrm; f2 UV; };
RT_DEV Attrib interpAttrib(const Tri3& v, float b1, float b2) {   // :249-271
  const float w0 = 1.0f - (b1 + b2);
  Attrib a;
  a.Pos = (w0 * v.pos[0] + b1 * v.pos[1]) + b2 * v.pos[2];
  a.Nrm = (w0 * v.nrm[0] + b1 * v.nrm[1]) + b2 * v.nrm[2];
  a.UV = getUV(a.Nrm, a.Pos, mk3(1.0f, 0.2f, 1.0f));
  return a;
}
RT_DEV f2 calcBarycentrics(const f4 p[3], f2 ndc) {   // :204-225
  const f3 invW = mk3(1.0f / p[0].w, 1.0f / p[1].w, 1.0f / p[2].w);
  f2 ndc0, ndc1, ndc2;
  ndc0.x = p[0].x * invW.x; ndc0.y = p[0].y * invW.x; ndc1.x = p[1].x * invW.y; ndc1.y = p[1].y * invW.y; ndc2.x = p[2].x * invW.z; ndc2.y = p[2].y * invW.z;
  const float det = (ndc2.x - ndc1.x) * (ndc0.y - ndc1.y) - (ndc2.y - ndc1.y) * (ndc0.x - ndc1.x);
  const float invDet = 1.0f / det;
  const f3 dPdx = mk3(ndc1.y - ndc2.y, ndc2.y - ndc0.y, ndc0.y - ndc1.y) * invDet;
  const f3 dPdy = mk3(ndc2.x - ndc1.x, ndc0.x - ndc2.x, ndc1.x - ndc0.x) * invDet;
  f2 dv; dv.x = ndc.x - ndc0.x; dv.y = ndc.y - ndc0.y;
  const float interpInvW = (invW.x + dv.x * dot3(invW, dPdx)) + dv.y * dot3(invW, dPdy);
  const float interpW = 1.0f / interpInvW;
  f2 b;
  b.x = interpW * (dv.x * dPdx.y * invW.y + dv.y * dPdy.y * invW.y);
  b.y = interpW * (dv.x * dPdx.z * invW.z + dv.y * dPdy.z * invW.z);
  return b;
}

// =========================================================================================================
// Kernel 1: ray generation
// =========================================================================================================
struct GenArgs {
  const unsigned long long* visDepth; uint32_t* depthOut;
  // the head of the NEXT frame's visibility pass (visibility.hip): its target is cleared and its lists are emptied here, on the way
  // (visNext == null: not).  One kernel launch per frame less, and the clear's 8 bytes per pixel ride on a kernel that is there anyway.
  // (Tried with it and not kept: merging the queued large triangles -- rasterLarge's work -- here as well.  Same arithmetic, but in
  // front of every wave's first dependent fetch: rayGenKernel 85 -> 169 us per launch, the 1080p frame 0.190 -> 0.243 ms, 4K 0.715 ->
  // 0.980; with the records in LDS no different.  profiles/r03_b_visibility_merge.txt)
  unsigned long long* visNext; uint32_t* zeroNext0; uint32_t* zeroNext1;
  // one word per tile of this kernel (rtggx_context.h visDirtyBuf): 0 = the tile of the target holds the clear value only.  visDirty: of the
  // target read here; visDirtyNext: of visNext; where the words are not known both point at words that are all ones (read / clear every
  // tile).  The words of visNext's tiles end as 0 (visDirtyNextOut: the target's own words)
  const uint32_t* visDirty; const uint32_t* visDirtyNext; uint32_t* visDirtyNextOut;
  uint32_t* normalOut; uint16_t* roughMetalOut; uint32_t* velocityOut; uint32_t* reflOut; uint32_t* diffOut;
  const uint16_t* roughMetalPrev;   // the previous frame's input set = what this target held before this frame
  const uint32_t* diffPrev;         // likewise RayTracingOut1, or null: the hit shading carries it over (launchShade)
  const float4* fat0; const float4* fat1;
  const uint2* env; const uint32_t* envMipOffset; uint32_t envSize, envMips;
  const float* cosSin;
  RayRec* rays; HitKey* hits; uint32_t* binCount; uint32_t binSlots;      // slots per bin (rt_queue.h)
  uint32_t* frameRays;      // 256 per-frame ray counters, zeroed here, added to by the trace kernel
  uint32_t tilesX, numTiles, rowBegin, rowEnd;
  // adaptive split (trace.hip): null / 0 when off
  uint32_t* binWork; uint32_t* splitList; uint32_t* splitCount; uint32_t splitWork, frontWork, splitMaxShift, splitCap;
};

#ifndef RT_GEN_MIN_BLOCKS
#define RT_GEN_MIN_BLOCKS 1
#endif
__global__ void __launch_bounds__(256, RT_GEN_MIN_BLOCKS) rayGenKernel(const FrameParams* __restrict__ fpp, GenArgs A) {
  const FrameParams& fp = *fpp;
  if (blockIdx.x == 0) A.frameRays[threadIdx.x] = 0u;
  if (blockIdx.x == 0 && threadIdx.x == 0 && A.visNext != nullptr) { *A.zeroNext0 = 0u; *A.zeroNext1 = 0u; }      // the next frame's large-triangle list, the next set's split list
  // 16x16 pixel tile per workgroup, 8x8 per wave: the 64 rays a wave appends are neighbours on screen
  const uint32_t tile = blockIdx.x;
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const uint32_t px = (tile % A.tilesX) * 16 + (wave & 1u) * 8 + (lane & 7u);
  const uint32_t py = A.rowBegin + (tile / A.tilesX) * 16 + (wave >> 1) * 8 + (lane >> 3);
  const bool inside = px < fp.W && py < A.rowEnd;
  // (two scalar loads, one wait: as two vector loads in front of the kernel's first fetch they cost ray generation 10 us in the frame)
  uint32_t wordHere, wordNext;
  asm volatile("s_load_dword %0, %2, %4\n\ts_load_dword %1, %3, %4\n\ts_waitcnt lgkmcnt(0)" : "=&s"(wordHere), "=&s"(wordNext) : "s"(A.visDirty), "s"(A.visDirtyNext), "s"(tile * 4u) : "memory");
  const bool drawn = wordHere != 0u;                                    // uniform over the workgroup
  const bool clearNext = A.visNext != nullptr && wordNext != 0u;
  const EnvRef env{A.env, A.envSize, A.envMips, A.envMipOffset};
  bool wantRefl = false, wantDiff = false;
  RayRec rr, rd;
  if (inside) {
    const uint32_t W = fp.W, H = fp.H;
    const size_t pix = (size_t)py * W + px;
    // getPrimarySurface :277-333
    const unsigned long long visWord = drawn ? A.visDepth[pix] : RT_VIS_CLEAR;
    if (clearNext) A.visNext[pix] = RT_VIS_CLEAR;
    // getSampleParam :394-406 -- of every pixel of a tile with something in it, before the visibility word is back: the table fetch then
    // travels beside that word instead of behind the triangle's
    float xiY = 0.0f, cosPhi = 0.0f, sinPhi = 0.0f;
    if (drawn) {
      uint32_t s = py * W + px;
      s = rng(s); s += fp.g.FrameIndex; s = rng(s); s %= 256u;
      xiY = (float)(rng(s) & 0xffffu) / 65536.0f;
      cosPhi = A.cosSin[s]; sinPhi = A.cosSin[256 + s];
    }
    uint32_t visibility = (uint32_t)visWord;
    A.depthOut[pix] = (uint32_t)(visWord >> 32);      // the filters read depth four bytes at a time instead of every other word of an 8-byte array
    f2 screenPos; screenPos.x = ((float)px + 0.5f) / (float)W * 2.0f - 1.0f; screenPos.y = ((float)py + 0.5f) / (float)H * 2.0f - 1.0f;
    screenPos.y = -screenPos.y;
    const f3 eye = mk3(fp.rg.EyePt[0], fp.rg.EyePt[1], fp.rg.EyePt[2]);
    bool hit; f3 N, V, P, color; f2 rghMtl, velocity; uint32_t inst = 0, prim = 0;
    if (visibility > 0) {
      --visibility;
      hit = true; inst = visibility >> 24; prim = visibility & 0xFFFFFFu;
      asm volatile("" : "+v"(prim));      // see shadeKernel: the mask must survive the array indexing by inst below
      const Tri3 v = getVertices(inst ? A.fat1 : A.fat0, prim);
      const M4 wvp = cbLoad4x4(fp.g.WorldViewProjs[inst]);
      f4 p[3];
      for (int k = 0; k < 3; ++k) p[k] = mulPoint(v.pos[k], wvp);
      screenPos.x -= fp.rg.ProjBias[0]; screenPos.y -= fp.rg.ProjBias[1];
      const f2 bary = calcBarycentrics(p, screenPos);
      const Attrib a = interpAttrib(v, bary.x, bary.y);
      color = mk3(fp.mat.BaseColors[inst][0], fp.mat.BaseColors[inst][1], fp.mat.BaseColors[inst][2]);
      rghMtl = getRoughMetal(fp.mat, inst, a.UV);
      const f4 hPrev = mulPoint(a.Pos, cbLoad4x4(fp.g.WorldViewProjsPrev[inst]));
      velocity.x = (screenPos.x - hPrev.x / hPrev.w) * 0.5f; velocity.y = (screenPos.y - hPrev.y / hPrev.w) * -0.5f;
      const f4 P4 = mulPoint(a.Pos, cbLoad4x3(fp.g.Worlds[inst]));
      P = mk3(P4.x, P4.y, P4.z);
      N = normalize3(mulDir(a.Nrm, cbLoad3x3(inst ? fp.g.WorldIT1 : fp.g.WorldITs0)));
      V = normalize3(eye - P);
    } else {
      f4 sp4; sp4.x = screenPos.x; sp4.y = screenPos.y; sp4.z = 0.0f; sp4.w = 1.0f;
      const f4 world = mulVec4(sp4, cbLoad4x4(fp.rg.ProjToWorld));
      hit = false; velocity.x = 0.0f; velocity.y = 0.0f;
      P = mk3(world.x / world.w, world.y / world.w, world.z / world.w);
      N = mk3(0.0f, 0.0f, 0.0f);
      V = normalize3(eye - P);
      rghMtl.x = 0.0f; rghMtl.y = 0.0f;
      color = mk3(0.0f, 0.0f, 0.0f);
    }
    // G-buffer stores :552-554
    A.normalOut[pix] = packR10G10B10A2(N.x * 0.5f + 0.5f, N.y * 0.5f + 0.5f, N.z * 0.5f + 0.5f, hit ? 1.0f : 0.0f);
    // the reference leaves RoughMetal untouched where nothing is hit, and RayTracingOut1 where no diffuse ray is traced:
    // with several input sets "untouched" means carrying the word of the previous frame's set over.  RoughMetal is carried here
    // (the previous set's was written by the previous ray generation, earlier on this stream); RayTracingOut1 below or by shadeKernel
    A.roughMetalOut[pix] = hit ? (uint16_t)packR8G8(rghMtl.x, rghMtl.y) : A.roughMetalPrev[pix];
    A.velocityOut[pix] = packR16G16F(velocity.x, velocity.y);

    if (!hit) {
      // degenerate ray [0,0] along -V always misses: missMain, environment mip 0; metallic 0 < 1 -> same for the diffuse target
      const uint32_t c = packR11G11B10F(environmentLevel0(env, -V));
      A.reflOut[pix] = c;
      A.diffOut[pix] = c;
    } else {
      const uint32_t skip = (inst << 24) | prim;
      {  // computeReflection depth 0 :424-484
        const float a = rghMtl.x * rghMtl.x;
        const bool vndf = (fp.flags & RT_FLAG_VNDF) != 0u;      // uniform
        f3 Hh;
        if (vndf) Hh = vndfHalfVector(N, V, a, cosPhi, sinPhi, xiY);
        else {
          const float cosTheta = sqrtf((1.0f - xiY) / (1.0f + (a * a - 1.0f) * xiY));
          const float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
          Hh = localToWorld(N, mk3(cosPhi * sinTheta, sinPhi * sinTheta, cosTheta));
        }
        const f3 R = reflect3(-V, Hh);
        const float NoL = dot3(N, R);
        if (NoL <= 0.0f) A.reflOut[pix] = 0u;   // :459
        else {
          const f3 f0 = mk3(lerpf(0.04f, color.x, rghMtl.y), lerpf(0.04f, color.y, rghMtl.y), lerpf(0.04f, color.z, rghMtl.y));
          const float NoV = saturatef(dot3(N, V));
          const float VoH = saturatef(dot3(V, Hh));
          const f3 F = fSchlick(f0, VoH);
          const float vis = visSmith(rghMtl.x, NoV, NoL);
          const float NoH = saturatef(dot3(N, Hh));
          const float k = 4.0f * VoH / NoH;
          wantRefl = true;
          rr.ox = P.x; rr.oy = P.y; rr.oz = P.z;
          rr.dx = R.x; rr.dy = R.y; rr.dz = R.z;
          rr.pixel = (uint32_t)pix; rr.skip = skip; rr.flags = 0u;
          rr.wx = ((NoL * F.x) * vis) * k; rr.wy = ((NoL * F.y) * vis) * k; rr.wz = ((NoL * F.z) * vis) * k;   // :477
          if (vndf) {      // BRDF x NoL / pdf of the visible-normal sampler = F x G2 / G1(V) = F x G1(L), with the separable Smith terms of Vis_Smith
            const float a2 = a * a;
            const float g1l = (2.0f * NoL) / (NoL + sqrtf(NoL * (NoL - NoL * a2) + a2));
            rr.wx = F.x * g1l; rr.wy = F.y * g1l; rr.wz = F.z * g1l;
          }
        }
      }
      if (rghMtl.y < 1.0f) {   // :559-564, computeDiffuse depth 0 :486-535
        const float cosTheta = 1.0f - 2.0f * xiY;
        const float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
        const f3 dir = normalize3(N + mk3(cosPhi * sinTheta, sinPhi * sinTheta, cosTheta));
        wantDiff = true;
        rd.ox = P.x; rd.oy = P.y; rd.oz = P.z;
        rd.dx = dir.x; rd.dy = dir.y; rd.dz = dir.z;
        rd.pixel = (uint32_t)pix; rd.skip = skip; rd.flags = 1u;
        rd.wx = color.x * (1.0f - 0.04f); rd.wy = color.y * (1.0f - 0.04f); rd.wz = color.z * (1.0f - 0.04f);   // :532
      } else if (A.diffPrev != nullptr) A.diffOut[pix] = A.diffPrev[pix];      // RayTracingOut1 keeps what it held: carried over from the previous frame's set, here or by shadeKernel (launchShade)
    }
  }
  if (clearNext && threadIdx.x == 0) A.visDirtyNextOut[tile] = 0u;      // (read above by this workgroup only)

  // wave-level compaction into this wave's own bin (rt_queue.h): reflection rays first, then diffuse rays
  const uint32_t bin = blockIdx.x * 4u + wave;
  const unsigned long long maskR = __ballot(wantRefl), maskD = __ballot(wantDiff), below = (1ull << lane) - 1ull;
  const uint32_t nR = (uint32_t)__popcll(maskR);
  RayRec* dst = A.rays + (size_t)bin * A.binSlots;
  HitKey* keys = A.hits + (size_t)bin * A.binSlots;      // every ray starts as a miss at TMax
  if (wantRefl) { const uint32_t k = (uint32_t)__popcll(maskR & below); dst[k] = rr; keys[k] = hitKey(RT_RAY_TMAX, 0xFFFFFFFFu); }
  // (a diffuse ray needs bins of two rays per pixel: rtggx_update_frame grows them with the first metallic below 1, and launchRayTrace
  // refuses a frame whose materials and bins disagree.  The bound here is the last line: a record is never written beyond its bin.)
  const uint32_t kD = nR + (uint32_t)__popcll(maskD & below);
  if (wantDiff && kD < A.binSlots) { dst[kD] = rd; keys[kD] = hitKey(RT_RAY_TMAX, 0xFFFFFFFFu); }
  const uint32_t nRaysInBin = min(nR + (uint32_t)__popcll(maskD), A.binSlots);
  if (A.binWork == nullptr) {
    if (lane == 0) A.binCount[bin] = nRaysInBin;
    return;
  }
  // Adaptive split (trace.hip).  What this bin's rays cost in the previous frame decides whether it goes on the split
  // list -- whose bins the trace kernel starts first -- and how many waves trace it: one per `splitWork` lane-steps, up
  // to 2^splitMaxShift.  One atomic per workgroup allocates the entries of its four bins; the count keeps running past
  // the list's capacity (the host sizes the next launches from it).
  __shared__ uint32_t want[4], listBase;
  uint32_t shift = 0u, n = 0u;
  if (lane == 0) {
    const uint32_t w = A.binWork[bin];
    A.binWork[bin] = 0u;
    while (shift < A.splitMaxShift && (w >> shift) > A.splitWork) ++shift;
    n = (shift || w > A.frontWork) ? 1u << shift : 0u;
    want[wave] = n;
  }
  __syncthreads();
  if (threadIdx.x == 0) { const uint32_t total = want[0] + want[1] + want[2] + want[3]; listBase = total ? atomicAdd(A.splitCount, total) : 0u; }
  __syncthreads();
  if (lane == 0) {
    uint32_t base = listBase, mark = 0u;
    for (uint32_t k = 0; k < wave; ++k) base += want[k];
    if (n) {
      const bool fits = base + n <= A.splitCap;
      for (uint32_t k = 0; k < n && base + k < A.splitCap; ++k) A.splitList[base + k] = fits ? ((shift << 28) | (k << 24) | bin) : 0xFFFFFFFFu;
      mark = fits ? (shift << 1) | 1u : 0u;
    }
    A.binCount[bin] = nRaysInBin | (mark << 8);     // rays in the bin | bit 8: on the split list | bits 9..: log2(its waves)
  }
}

// =========================================================================================================
// Kernel 3: hit / miss shading
// =========================================================================================================
struct ShadeArgs {
  const RayRec* rays; const HitKey* hits; const uint32_t* binCount; uint32_t binSlots;
  const float4* fat0; const float4* fat1;
  const uint2* env; const uint32_t* envMipOffset; uint32_t envSize, envMips;
  const float* sh;
  uint32_t* reflOut; uint32_t* diffOut;
  // carry-over of RayTracingOut1 (see the kernel)
  const uint32_t* diffPrev; const unsigned long long* visDepth; uint32_t tilesX, rowBegin, rowEnd, carryMask;
  const uint32_t* tileWords;      // one word per tile of this kernel, 0 = nothing was drawn there (rtggx_context.h visDirtyBuf): no rays, nothing to carry
};

// computeReflection at recursion depth 1 (:424-484)
RT_DEV f3 reflectionDepth1(const EnvRef& env, f2 rghMtl, f3 N, f3 V, f3 color) {
  const float level = calcMipFromRoughness(rghMtl.x, (float)env.mips);
  const float a = rghMtl.x * rghMtl.x;
  const f3 R = reflect3(-V, N);
  const f3 dir = lerp3(N, R, (1.0f - a) * (sqrtf(1.0f - a) + a));
  const float NoL = dot3(N, dir);
  if (NoL <= 0.0f) return mk3(0.0f, 0.0f, 0.0f);
  const f3 e = environment(env, dir, level);
  const f3 f0 = mk3(lerpf(0.04f, color.x, rghMtl.y), lerpf(0.04f, color.y, rghMtl.y), lerpf(0.04f, color.z, rghMtl.y));
  const float NoV = saturatef(dot3(N, V));
  return e * envBRDFApprox(f0, rghMtl.x, NoV);
}

__global__ void __launch_bounds__(256, RT_GEN_MIN_BLOCKS) shadeKernel(const FrameParams* __restrict__ fpp, ShadeArgs A) {
  const FrameParams& fp = *fpp;
  const EnvRef env{A.env, A.envSize, A.envMips, A.envMipOffset};
  // workgroup b shades the four bins its rayGen namesake filled: wave w <-> bin 4b + w
  // (Round 4, measured and dropped: a LIST of the bins -- or tiles -- with a surface, appended by ray generation, walked here by a grid of
  // eight workgroups per CU, instead of three quarters of this kernel's waves reading a zero and leaving.  One atomic per wave with a
  // surface, ~8 000 per frame on one word, runs at the memory side of eight L2s: ray generation 80 -> 155 us in the frame, the frame
  // 0.185 -> 0.266 ms; one per tile behind a workgroup barrier: ray generation 80 -> 91 us, the frame +1.3 %.  profiles/r04_j_tile_words.txt)
  const uint32_t tile = blockIdx.x;
  { uint32_t word;      // three quarters of the bunny frame's workgroups leave here, after one scalar load (before: a vector load of the bin's count each)
    asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(word) : "s"(A.tileWords), "s"(tile * 4u) : "memory");
    if (word == 0u) return; }
  const uint32_t bin = tile * 4u + (threadIdx.x >> 6);
  // Carry-over of RayTracingOut1.  The reference has ONE such texture and leaves it untouched where no diffuse ray is traced
  // (covered pixels of a fully metallic instance, RayTracing.hlsl:559): it keeps the last value ever written there.  With three
  // input sets that is the word of the previous frame's set -- final only once that frame's shading has run, which is earlier
  // on THIS stream; ray generation (stream C, a frame ahead of this stream) must not read it.  The wave's bin is its 8x8 pixel
  // sub-tile, so it walks those pixels: covered by an instance with metallic >= 1 (carryMask bit per instance) -> copy.
  if (A.carryMask != 0u) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t px = (tile % A.tilesX) * 16u + (wave & 1u) * 8u + (lane & 7u);
    const uint32_t py = A.rowBegin + (tile / A.tilesX) * 16u + (wave >> 1) * 8u + (lane >> 3);
    if (px < fp.W && py < A.rowEnd) {
      const size_t pix = (size_t)py * fp.W + px;
      const uint32_t vis = (uint32_t)A.visDepth[pix];
      if (vis != 0u && ((A.carryMask >> ((vis - 1u) >> 24)) & 1u)) A.diffOut[pix] = A.diffPrev[pix];
    }
  }
  const uint32_t count = min(A.binCount[bin] & 0xFFu, A.binSlots);
  for (uint32_t i = threadIdx.x & 63u; i < count; i += 64u) {
    const size_t slot = (size_t)bin * A.binSlots + i;
    const float4* rp = reinterpret_cast<const float4*>(A.rays + slot);
    const float4 q0 = rp[0], q1 = rp[1], q2 = rp[2];
    // the round-1 names: ra = origin, rb = direction, rc = (pixel, skip, flags), rw = weight
    const float4 ra = make_float4(q0.x, q0.y, q0.z, RT_RAY_TMIN), rb = make_float4(q0.w, q1.x, q1.y, RT_RAY_TMAX), rw = make_float4(q2.x, q2.y, q2.z, 0.0f);
    const uint4 rc = make_uint4(__float_as_uint(q1.z), __float_as_uint(q1.w), __float_as_uint(q2.w), 0u);
    const uint32_t hitId = hitKeyId(A.hits[slot]);
    const f3 dir = mk3(rb.x, rb.y, rb.z);
    const bool diffuseGroup = (rc.z & 1u) != 0u;
    const uint32_t srcInst = rc.y >> 24;
    f3 col;
    if (hitId == 0xFFFFFFFFu) col = environmentLevel0(env, dir);   // missMain :620-625
    else {
      // payload preset = color * metallic of the surface the ray left (:456); closestHitReflection returns it untouched when <= 0 (:573)
      const float m = fp.mat.RoughMetals[srcInst][1];
      const f3 preset = mk3(fp.mat.BaseColors[srcInst][0] * m, fp.mat.BaseColors[srcInst][1] * m, fp.mat.BaseColors[srcInst][2] * m);
      if (!diffuseGroup && preset.x <= 0.0f && preset.y <= 0.0f && preset.z <= 0.0f) col = preset;
      else {
        const uint32_t hInst = hitId >> 24;
        uint32_t hPrim = hitId & 0xFFFFFFu;
        // hipcc 7.2 drops this mask when the same id also indexes a two-element array (it then addresses the triangle
        // with the whole id: a fault for every hit on instance 1): keep it behind a barrier, and select instead of index
        asm volatile("" : "+v"(hPrim));
        const Tri3 v = getVertices(hInst ? A.fat1 : A.fat0, hPrim);
        // the hit attributes (barycentrics) of the recorded triangle: the traversal's own test, repeated
        float ht, hb1 = 0.0f, hb2 = 0.0f;
        woopTestVerts(toObject(ra.x, ra.y, ra.z, rb.x, rb.y, rb.z, hInst ? fp.invWorld[1] : fp.invWorld[0]), v.pos[0], v.pos[1], v.pos[2], ht, hb1, hb2);
        const Attrib a = interpAttrib(v, hb1, hb2);
        const f3 N = normalize3(mulDir(a.Nrm, cbLoad3x3(hInst ? fp.g.WorldIT1 : fp.g.WorldITs0)));
        const f2 rm = getRoughMetal(fp.mat, hInst, a.UV);
        f3 color = mk3(fp.mat.BaseColors[hInst][0], fp.mat.BaseColors[hInst][1], fp.mat.BaseColors[hInst][2]);
        const f3 V = -dir;
        if (rm.y > 0.5f) col = reflectionDepth1(env, rm, N, V, color);
        else {
          if (diffuseGroup) color = color * (1.0f - rm.y);                 // :607
          const f3 irr = evaluateSHIrradiance(A.sh, N);                    // computeDiffuse depth 1 :513,532
          col = mk3(irr.x / RT_PI, irr.y / RT_PI, irr.z / RT_PI) * color;
        }
      }
    }
    const uint32_t packed = packR11G11B10F(mk3(col.x * rw.x, col.y * rw.y, col.z * rw.z));
    if (diffuseGroup) A.diffOut[rc.x] = packed; else A.reflOut[rc.x] = packed;
  }
}

// =========================================================================================================
// host side
// =========================================================================================================
int launchRayTrace(rtggx_context* c, const FrameParams& fp, hipStream_t sGen, hipStream_t s, hipEvent_t done) {
  uint32_t rb, re;
  passRows(fp, ROWS_GBUFFER, rb, re);
  if (re <= rb) return 0;
  const uint32_t tilesX = (fp.W + 15) / 16, tilesY = (re - rb + 15) / 16;
  if ((fp.mat.RoughMetals[0][1] < 1.0f || fp.mat.RoughMetals[1][1] < 1.0f) && c->binSlots < RT_BIN) {
    setError("rtggx_ray_trace: a material with metallic below 1 (a diffuse ray per pixel as well) but ray bins of %u slots: rtggx_update_frame sizes them", c->binSlots); return -1;
  }
  GenArgs G;
  // ray generation starts the visibility pass of the frame after next (GenArgs): its target cleared, its list of large triangles emptied
  // (the list by frame parity: the one this frame's visibility pass has just used up), and the next set's split list
  { const uint32_t thenFrame = c->frameCounter + 2u, nextSet = (c->setIndex + 1u) % RT_SETS;
    G.visNext = c->visDepthBuf[thenFrame % RT_VIS_RING]; G.zeroNext0 = c->largeCountBase + (thenFrame & 1u); G.zeroNext1 = c->largeCountBase + 2u + nextSet;
    auto& vc = c->visClearedAt[thenFrame % RT_VIS_RING]; vc.frame = thenFrame; vc.rows[0] = rb; vc.rows[1] = re;
    // the tiles' words (rtggx_context.h visDirtyBuf): usable where they were kept for these very rows
    if (c->traceShare > 0.93f) c->traversalBound = true; else if (c->traceShare < 0.89f) c->traversalBound = false;
    G.visDirty = c->traceTileWords = c->tileWords(rb, re);
    auto& vn = c->visFlags[thenFrame % RT_VIS_RING];
    G.visDirtyNextOut = c->visDirtyBuf[thenFrame % RT_VIS_RING];
    G.visDirtyNext = c->useTileWords && !c->traversalBound && vn.rows[0] == rb && vn.rows[1] == re ? G.visDirtyNextOut : c->visDirtyOnes;
    vn.rows[0] = rb; vn.rows[1] = re;
  }
  G.visDepth = c->visDepth; G.depthOut = c->depth32; G.normalOut = c->normal; G.roughMetalOut = c->roughMetal; G.velocityOut = c->velocity; G.reflOut = c->rtRefl; G.diffOut = c->rtDiff;
  G.roughMetalPrev = c->roughMetalBuf[(c->setIndex + RT_SETS - 1u) % RT_SETS];   // the previous frame's set
  G.diffPrev = c->genCarriesDiff ? c->rtDiffBuf[(c->setIndex + RT_SETS - 1u) % RT_SETS] : nullptr;
  G.fat0 = c->mesh[0].fat; G.fat1 = c->mesh[1].fat;
  G.env = c->env.texels; G.envMipOffset = c->dEnvMipOffset; G.envSize = c->env.size; G.envMips = c->env.mips; G.cosSin = c->cosSinTab;
  G.rays = (RayRec*)c->rayQueue; G.hits = (HitKey*)c->hitQueue; G.binCount = c->binCount; G.binSlots = c->binSlots; G.frameRays = c->rayCounter32;
  G.tilesX = tilesX; G.numTiles = tilesX * tilesY; G.rowBegin = rb; G.rowEnd = re;
  const uint32_t splitWork = c->splitWork, splitMaxShift = c->splitMaxShift;
  const uint32_t sliceShift = chooseSliceShift(c, true, G.numTiles * 4u);
  // "wide" launches: few enough rays that the traversal does not fill the chip for long (trace.hip launchTrace, capi.hip rtggx_ray_trace)
  c->lastTraceSmall = c->forcePlacement >= 0 ? c->forcePlacement == 1 : (sliceShift > 0u || c->lastFrameRays < RT_WIDE_RAYS);
  const bool adaptive = splitWork != 0u && sliceShift == 0u;
  c->lastTraceAdaptive = adaptive;
  // the split list is sized from the demand of an earlier frame (copied back asynchronously, like the ray counters)
  const uint32_t splitCap = !adaptive ? 0u : c->splitCapForced != 0xFFFFFFFFu ? c->splitCapForced
                          : c->splitDemand == 0u ? 0u : ((c->splitDemand + c->splitDemand / 8u + 64u + 31u) / 32u) * 32u;
  G.binWork = adaptive ? c->binWork : nullptr; G.splitList = c->splitList; G.splitCount = c->splitCount;
  G.frontWork = RT_SPLIT_FRONT < splitWork ? RT_SPLIT_FRONT : splitWork;
  G.splitWork = splitWork; G.splitMaxShift = splitMaxShift < 3u ? splitMaxShift : 3u; G.splitCap = splitCap < RT_SPLIT_CAP ? splitCap : RT_SPLIT_CAP;
  const hipEvent_t evGen = c->evGenRing[c->frameCounter & 3u];
  if (sGen != s && c->attachEvents) hipExtLaunchKernelGGL(rayGenKernel, dim3(G.numTiles), dim3(256), 0, sGen, nullptr, evGen, 0, (const FrameParams*)(c->dParams + c->slot), G);
  else hipLaunchKernelGGL(rayGenKernel, dim3(G.numTiles), dim3(256), 0, sGen, c->dParams + c->slot, G);
  c->genFrame[c->frameCounter & 3u] = 0u; c->genStreamOf[c->frameCounter & 3u] = sGen;
  if (sGen != s) {      // ray generation on stream C, the traversal on stream B behind it
    if (!c->attachEvents) RT_HIP(hipEventRecord(evGen, sGen));
    RT_HIP(hipStreamWaitEvent(s, evGen, 0));
    c->genFrame[c->frameCounter & 3u] = c->frameCounter;      // (the event exists: a visibility pass on another stream two frames on waits for it)
  }
  if (c->timing) hipEventRecord(c->tev[11], s);
  const bool ring = c->kernelRing && c->kevCount < c->kevBegin.size() && (c->ringTick++ % c->ringStride) == 0u;
  // a sampled frame: the event pair of the kernel ring rides on the dispatch (and `done` is recorded behind it)
  const bool attach = c->attachEvents;
  if (ring && !attach) hipEventRecord(c->kevBegin[c->kevCount], s);
  { const int r = launchTrace(c, fp, s, G.numTiles * 4u, true, tilesX, tilesY, sliceShift, adaptive ? (int)G.splitCap : -1,
                              ring && attach ? c->kevBegin[c->kevCount] : nullptr, !attach ? nullptr : ring ? c->kevEnd[c->kevCount] : done); if (r) return r; }
  if (c->timing) hipEventRecord(c->tev[12], s);
  if (ring && !attach) hipEventRecord(c->kevEnd[c->kevCount], s);
  if (ring) ++c->kevCount;
  if (done && (ring || !attach)) hipEventRecord(done, s);
  RT_HIP(hipGetLastError());
  return 0;
}

int launchShade(rtggx_context* c, const FrameParams& fp, hipStream_t s, hipEvent_t done) {
  uint32_t rb, re;
  passRows(fp, ROWS_GBUFFER, rb, re);
  if (re <= rb) return 0;
  const uint32_t tilesX = (fp.W + 15) / 16, numTiles = tilesX * ((re - rb + 15) / 16);
  ShadeArgs S;
  S.diffPrev = c->rtDiffBuf[(c->setIndex + RT_SETS - 1u) % RT_SETS]; S.visDepth = c->visDepth; S.tilesX = tilesX; S.rowBegin = rb; S.rowEnd = re;
  S.carryMask = (fp.mat.RoughMetals[0][1] >= 1.0f ? 1u : 0u) | (fp.mat.RoughMetals[1][1] >= 1.0f ? 2u : 0u);      // rghMtl.y < 1 is the test of :559; it is the instance's constant
  // ... unless ray generation has carried those pixels over already.  It can when the previous frame's shading kernel wrote nothing
  // into the previous set's image (no diffuse rays, no carrying): that image was final when the previous ray generation ended, earlier
  // on the same stream.  The steady state of an all-metal scene: 8 bytes per covered pixel less (this loop reads the visibility word
  // again), and no dependency between the shading kernels of consecutive frames (capi.hip rtggx_ray_trace).
  if (c->genCarriesDiff) S.carryMask = 0u;
  S.tileWords = c->tileWords(rb, re);
  S.rays = (const RayRec*)c->rayQueue; S.hits = (const HitKey*)c->hitQueue; S.binCount = c->binCount; S.binSlots = c->binSlots;
  S.fat0 = c->mesh[0].fat; S.fat1 = c->mesh[1].fat;
  S.env = c->env.texels; S.envMipOffset = c->dEnvMipOffset; S.envSize = c->env.size; S.envMips = c->env.mips; S.sh = c->sh;
  S.reflOut = c->rtRefl; S.diffOut = c->rtDiff;
  if (done && c->attachEvents) hipExtLaunchKernelGGL(shadeKernel, dim3(numTiles), dim3(256), 0, s, nullptr, done, 0, (const FrameParams*)(c->dParams + c->slot), S);
  else {
    hipLaunchKernelGGL(shadeKernel, dim3(numTiles), dim3(256), 0, s, c->dParams + c->slot, S);
    if (done) hipEventRecord(done, s);
  }
  RT_HIP(hipGetLastError());
  return 0;
}

// ---- test entry: closest-hit queries for an explicit ray list (through the same trace kernel) ---------------
__global__ void fillTestQueue(const float* __restrict__ rays, uint32_t n, uint32_t binSlots, RayRec* q0, HitKey* keys, uint32_t* binCount, float2* tRange) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i % binSlots == 0 && i < n) binCount[i / binSlots] = n - i < binSlots ? n - i : binSlots;   // bins are filled densely, in order
  if (i >= n) return;
  const float* r = rays + 8 * (size_t)i;
  RayRec rr;
  rr.ox = r[0]; rr.oy = r[1]; rr.oz = r[2]; rr.dx = r[3]; rr.dy = r[4]; rr.dz = r[5];
  rr.pixel = 0u; rr.skip = 0xFFFFFFFFu; rr.flags = 0u; rr.wx = rr.wy = rr.wz = 0.0f;
  q0[i] = rr;
  tRange[i] = make_float2(r[6], r[7]);      // these rays bring their own interval
  keys[i] = hitKey(r[7], 0xFFFFFFFFu);
}
__global__ void exportTestHits(const FrameParams* __restrict__ fpp, const RayRec* __restrict__ rays, const HitKey* __restrict__ hits, uint32_t n,
                               const float4* __restrict__ fat0, const float4* __restrict__ fat1, float* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const HitKey k = hits[i];
  const uint32_t id = hitKeyId(k);
  float* o = out + 6 * (size_t)i;
  const bool valid = id != 0xFFFFFFFFu;
  float t = hitKeyT(k), b1 = 0.0f, b2 = 0.0f;
  if (valid) {
    const RayRec rr = rays[i];
    const uint32_t inst = id >> 24;
    uint32_t prim = id & 0xFFFFFFu;
    // (same compiler hazard as in shadeKernel: keep the mask behind a barrier)
    asm volatile("" : "+v"(prim));
    const Tri3 v = getVertices(inst ? fat1 : fat0, prim);
    woopTestVerts(toObject(rr.ox, rr.oy, rr.oz, rr.dx, rr.dy, rr.dz, inst ? fpp->invWorld[1] : fpp->invWorld[0]), v.pos[0], v.pos[1], v.pos[2], t, b1, b2);
  }
  o[0] = t; o[1] = u2f(valid ? id >> 24 : 0u); o[2] = u2f(valid ? id & 0xFFFFFFu : 0u); o[3] = b1; o[4] = b2; o[5] = valid ? 1.0f : 0.0f;
}
int launchTraceRays(rtggx_context* c, const FrameParams& fp, const float* dRays, uint32_t n, float* dOut, hipStream_t s) {
  if (!n) return 0;
  if (n > c->numBinsMax * c->binSlots) { setError("rtggx_trace_rays: at most %u rays per launch", c->numBinsMax * c->binSlots); return -1; }
  const uint32_t numBins = (((n + c->binSlots - 1u) / c->binSlots) + 3u) & ~3u;   // whole tiles of four bins
  RT_HIP(hipMemsetAsync(c->binCount, 0, (size_t)numBins * 4, s));
  // the rays' own (TMin, TMax): one float2 per slot, kept for the context's lifetime once a caller has used this entry point
  if (!c->testRayRange) RT_HIP(hipMalloc(&c->testRayRange, (size_t)c->numBinsMax * RT_BIN * sizeof(float2)));
  hipLaunchKernelGGL(fillTestQueue, dim3((n + 255) / 256), dim3(256), 0, s, dRays, n, c->binSlots, (RayRec*)c->rayQueue, (HitKey*)c->hitQueue, c->binCount, (float2*)c->testRayRange);
  c->traceRayRange = c->testRayRange;
  { const int r = launchTrace(c, fp, s, numBins, false, 0u, 0u, chooseSliceShift(c, false, numBins), -1); c->traceRayRange = nullptr; if (r) return r; }
  hipLaunchKernelGGL(exportTestHits, dim3((n + 255) / 256), dim3(256), 0, s, c->dParams + c->slot, (const RayRec*)c->rayQueue, (const HitKey*)c->hitQueue, n,
                     (const float4*)c->mesh[0].fat, (const float4*)c->mesh[1].fat, dOut);
  RT_HIP(hipGetLastError());
  return 0;
}

}  // namespace rt
