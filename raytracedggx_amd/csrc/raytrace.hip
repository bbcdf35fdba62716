// Ray-tracing pass: the gfx950 replacement of DispatchRays(W,H,1) over RayTracing.cso
// (RayTracer::rayTrace, RayTracedGGX/Content/RayTracer.cpp:793-810).  One kernel fuses what DXR
// spreads over four shaders of RayTracedGGX/Content/Shaders/RayTracing.hlsl:
//   raygenMain :541-565 (G-buffer reconstruction from the visibility buffer, GGX / cosine
//   sampling, BRDF weighting), closestHitReflection :571-590, closestHitDiffuse :593-614,
//   missMain :620-625, with Material.hlsli, BRDFModels.hlsli, SHIrradianceTypeless.hlsli:16-37.
//
// Traversal (TraceRay, RayTracing.hlsl:183-198): two-level software BVH.  Each lane owns one
// pixel and one ray; the ray is carried into each instance's object space (TLAS = two
// world->object matrices), then walks the 64-byte-node binary LBVH with a per-lane stack held in
// LDS ([entry][lane] layout: conflict-free), nearer child first.  Ray/triangle: watertight test
// (Woop, Benthin, Wald 2013), no culling, TMin < t < TMax, ties to the lower (instance, primitive).
// A wave covers an 8x8 pixel tile so that its 64 rays are coherent; blockIdx is remapped so that
// consecutive tiles of the screen stay on one XCD (its L2 then holds the BVH region they touch).
//
// Roofline: HBM by decree of the metric (no MFMA work exists here); algorithmic bytes per pixel:
// 4 (visibility) in, 4+2+4+4 out (+4 when metallic < 1) -- the BVH (<= 10 MB) is L2/MALL resident,
// so the kernel is latency/issue bound, not HBM bound (DESIGN.md "Roofline").
#include "rtggx_context.h"

namespace rt {

#define RT_STACK 48
#define RT_PI 3.1415926535897f   // BRDFModels.hlsli:5

struct Hit { float t; uint32_t inst, prim; float b1, b2; bool valid; };

struct RayX { float o[3], d[3], invd[3]; int kx, ky, kz; float Sx, Sy, Sz; };

RT_DEV float sel3(const float* v, int k) { return k == 0 ? v[0] : (k == 1 ? v[1] : v[2]); }

RT_DEV RayX rayToObject(f3 o, f3 d, const float* inv) {
  M4 M; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) M.m[i][j] = inv[i * 4 + j];
  RayX r;
  const f4 oo = mulPoint(o, M);
  const f3 dd = mulDir(d, M);
  r.o[0] = oo.x; r.o[1] = oo.y; r.o[2] = oo.z;
  r.d[0] = dd.x; r.d[1] = dd.y; r.d[2] = dd.z;
  for (int k = 0; k < 3; ++k) r.invd[k] = 1.0f / r.d[k];
  const float ax = fabsf(r.d[0]), ay = fabsf(r.d[1]), az = fabsf(r.d[2]);
  r.kz = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);
  r.kx = (r.kz + 1) % 3; r.ky = (r.kx + 1) % 3;
  if (sel3(r.d, r.kz) < 0.0f) { const int t = r.kx; r.kx = r.ky; r.ky = t; }
  const float dz = sel3(r.d, r.kz);
  r.Sx = sel3(r.d, r.kx) / dz; r.Sy = sel3(r.d, r.ky) / dz; r.Sz = 1.0f / dz;
  return r;
}

RT_DEV bool woopIntersect(const RayX& r, const BvhTri& tr, float& t, float& b1, float& b2) {
  const float A[3] = {tr.v0[0] - r.o[0], tr.v0[1] - r.o[1], tr.v0[2] - r.o[2]};
  const float B[3] = {tr.v1[0] - r.o[0], tr.v1[1] - r.o[1], tr.v1[2] - r.o[2]};
  const float C[3] = {tr.v2[0] - r.o[0], tr.v2[1] - r.o[1], tr.v2[2] - r.o[2]};
  const float Akz = sel3(A, r.kz), Bkz = sel3(B, r.kz), Ckz = sel3(C, r.kz);
  const float Ax = sel3(A, r.kx) - r.Sx * Akz, Ay = sel3(A, r.ky) - r.Sy * Akz;
  const float Bx = sel3(B, r.kx) - r.Sx * Bkz, By = sel3(B, r.ky) - r.Sy * Bkz;
  const float Cx = sel3(C, r.kx) - r.Sx * Ckz, Cy = sel3(C, r.ky) - r.Sy * Ckz;
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

RT_DEV void slab(const RayX& r, const float* bmin, const float* bmax, float tmin, float tmax, float& tn, float& tf) {
  const float x1 = (bmin[0] - r.o[0]) * r.invd[0], x2 = (bmax[0] - r.o[0]) * r.invd[0];
  const float y1 = (bmin[1] - r.o[1]) * r.invd[1], y2 = (bmax[1] - r.o[1]) * r.invd[1];
  const float z1 = (bmin[2] - r.o[2]) * r.invd[2], z2 = (bmax[2] - r.o[2]) * r.invd[2];
  tn = fmaxf(fmaxf(fminf(x1, x2), fminf(y1, y2)), fmaxf(fminf(z1, z2), tmin));
  tf = fminf(fminf(fmaxf(x1, x2), fmaxf(y1, y2)), fminf(fmaxf(z1, z2), tmax));
}

// Closest hit over both instances.  `stack` points at this lane's column of the LDS stack
// (stride = blockDim.x entries).
__device__ __noinline__ void traceClosest(const Scene& sc, const float* invWorld0, const float* invWorld1, f3 o, f3 d, float tmin, float tmax,
                                          uint32_t skipInst, uint32_t skipPrim, int32_t* stack, uint32_t stride, Hit& best) {
  best.t = tmax; best.inst = 0; best.prim = 0; best.b1 = 0.0f; best.b2 = 0.0f; best.valid = false;
  if (!(tmax > tmin)) return;
  for (uint32_t inst = 0; inst < 2; ++inst) {
    if (sc.tris[inst] == nullptr) continue;   // empty mesh (root is ~0 = -1 for a single-triangle mesh)
    const RayX r = rayToObject(o, d, inst ? invWorld1 : invWorld0);
    const BvhNode* __restrict__ nodes = sc.nodes[inst];
    const BvhTri* __restrict__ tris = sc.tris[inst];
    int sp = 0;
    int32_t cur = sc.root[inst];
    for (;;) {
      if (cur < 0) {
        const BvhTri tr = tris[~cur];
        if (!(inst == skipInst && tr.prim == skipPrim)) {
          float t, b1, b2;
          if (woopIntersect(r, tr, t, b1, b2) && t > tmin) {
            const bool closer = t < best.t;
            const bool tie = best.valid && t == best.t && (inst < best.inst || (inst == best.inst && tr.prim < best.prim));
            if (closer || tie) { best.t = t; best.inst = inst; best.prim = tr.prim; best.b1 = b1; best.b2 = b2; best.valid = true; }
          }
        }
        if (sp == 0) break;
        cur = stack[(--sp) * stride];
        continue;
      }
      const BvhNode nd = nodes[cur];
      float ln, lf, rn, rf;
      slab(r, nd.lmin, nd.lmax, tmin, best.t, ln, lf);
      slab(r, nd.rmin, nd.rmax, tmin, best.t, rn, rf);
      const bool hl = ln <= lf * 1.0000004f, hr = rn <= rf * 1.0000004f;
      if (hl && hr) {
        const bool leftFirst = ln <= rn;
        if (sp < RT_STACK) stack[(sp++) * stride] = leftFirst ? nd.right : nd.left;
        cur = leftFirst ? nd.left : nd.right;
      } else if (hl) cur = nd.left;
      else if (hr) cur = nd.right;
      else { if (sp == 0) break; cur = stack[(--sp) * stride]; }
    }
  }
}

// ---- environment (RayTracing.hlsl:167-180; D3D cube sampling restated, see oracle/orc_raytrace.h) ---
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
RT_DEV f3 cubeTexel(const Scene& sc, uint32_t mip, int face, int x, int y) {
  const int s = (int)((sc.envSize >> mip) ? (sc.envSize >> mip) : 1u);
  if (x < 0 || y < 0 || x >= s || y >= s) {
    const float u = ((float)x + 0.5f) / (float)s * 2.0f - 1.0f, v = ((float)y + 0.5f) / (float)s * 2.0f - 1.0f;
    float uu, vv; cubeFaceUV(cubeFaceDir(face, u, v), face, uu, vv);
    x = (int)floorf((uu * 0.5f + 0.5f) * (float)s); y = (int)floorf((vv * 0.5f + 0.5f) * (float)s);
    x = min(max(x, 0), s - 1); y = min(max(y, 0), s - 1);
  }
  const uint2 t = sc.env[sc.mipOffset[mip] + (uint32_t)face * (uint32_t)(s * s) + (uint32_t)(y * s + x)];
  return mk3(f16ToF32(t.x & 0xFFFFu), f16ToF32(t.x >> 16), f16ToF32(t.y & 0xFFFFu));
}
RT_DEV f3 cubeBilinear(const Scene& sc, uint32_t mip, int face, float u, float v) {
  const int s = (int)((sc.envSize >> mip) ? (sc.envSize >> mip) : 1u);
  const float x = (u * 0.5f + 0.5f) * (float)s - 0.5f, y = (v * 0.5f + 0.5f) * (float)s - 0.5f;
  const float x0 = floorf(x), y0 = floorf(y);
  const float fx = x - x0, fy = y - y0;
  const int ix = (int)x0, iy = (int)y0;
  const float w00 = (1.0f - fx) * (1.0f - fy), w10 = fx * (1.0f - fy), w01 = (1.0f - fx) * fy, w11 = fx * fy;
  const f3 c00 = cubeTexel(sc, mip, face, ix, iy), c10 = cubeTexel(sc, mip, face, ix + 1, iy);
  const f3 c01 = cubeTexel(sc, mip, face, ix, iy + 1), c11 = cubeTexel(sc, mip, face, ix + 1, iy + 1);
  return ((c00 * w00 + c10 * w10) + c01 * w01) + c11 * w11;
}
__device__ __noinline__ f3 environment(const Scene& sc, f3 dir, float level) {
  int face; float u, v; cubeFaceUV(dir, face, u, v);
  const float maxLevel = (float)(sc.envMips - 1);
  const float l = fminf(fmaxf(level, 0.0f), maxLevel);
  const float l0 = floorf(l), fl = l - l0;
  const uint32_t m0 = (uint32_t)l0, m1 = min(m0 + 1, sc.envMips - 1);
  const f3 a = cubeBilinear(sc, m0, face, u, v);
  if (fl == 0.0f) return a;
  const f3 b = cubeBilinear(sc, m1, face, u, v);
  return a * (1.0f - fl) + b * fl;
}

// ---- SHIrradianceTypeless.hlsli:16-37 ------------------------------------------------------------------
RT_DEV f3 evaluateSHIrradiance(const Scene& sc, f3 norm) {
  const float c1 = 0.42904276540489171563379376569857f, c2 = 0.51166335397324424423977581244463f;
  const float c3 = 0.24770795610037568833406429782001f, c4 = 0.88622692545275801364908374167057f;
  const float x = -norm.x, y = -norm.y, z = norm.z;
#define SHL(i) mk3(sc.sh[3 * (i)], sc.sh[3 * (i) + 1], sc.sh[3 * (i) + 2])
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
  const float a004 = fminf(rx * rx, exp2f(-9.28f * NoV)) * rx + ry;
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
  const float level = 3.0f - 1.15f * log2f(rgh);
  return mipCount - 1.0f - level;
}
RT_DEV f3 localToWorld(f3 n, f3 l) {   // :129-147
  const f3 up = fabsf(n.y) < 0.999f ? mk3(0.0f, 1.0f, 0.0f) : mk3(1.0f, 0.0f, 0.0f);
  const f3 xAxis = normalize3(cross3(up, n));
  const f3 yAxis = cross3(n, xAxis);
  return (xAxis * l.x + yAxis * l.y) + n * l.z;
}
struct Tri3 { f3 pos[3], nrm[3]; };
RT_DEV Tri3 getVertices(const Scene& sc, uint32_t inst, uint32_t prim) {   // :230-244
  Tri3 v;
  const uint32_t* idx = sc.idx[inst] + 3 * (size_t)prim;
  for (int k = 0; k < 3; ++k) {
    const float* p = sc.verts[inst] + 6 * (size_t)idx[k];
    v.pos[k] = mk3(p[0], p[1], p[2]); v.nrm[k] = mk3(p[3], p[4], p[5]);
  }
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

// computeReflection at recursion depth 1 (:424-484)
RT_DEV f3 reflectionDepth1(const Scene& sc, f2 rghMtl, f3 N, f3 V, f3 color) {
  const float level = calcMipFromRoughness(rghMtl.x, (float)sc.envMips);
  const float a = rghMtl.x * rghMtl.x;
  const f3 R = reflect3(-V, N);
  const f3 dir = lerp3(N, R, (1.0f - a) * (sqrtf(1.0f - a) + a));
  const float NoL = dot3(N, dir);
  if (NoL <= 0.0f) return mk3(0.0f, 0.0f, 0.0f);
  const f3 env = environment(sc, dir, level);
  const f3 f0 = mk3(lerpf(0.04f, color.x, rghMtl.y), lerpf(0.04f, color.y, rghMtl.y), lerpf(0.04f, color.z, rghMtl.y));
  const float NoV = saturatef(dot3(N, V));
  return env * envBRDFApprox(f0, rghMtl.x, NoV);
}
// computeDiffuse at recursion depth 1 (:486-535)
RT_DEV f3 diffuseDepth1(const Scene& sc, f3 N, f3 color) {
  const f3 irr = evaluateSHIrradiance(sc, N);
  return mk3(irr.x / RT_PI, irr.y / RT_PI, irr.z / RT_PI) * color;
}
// closestHitReflection :571-590 / closestHitDiffuse :593-614 (diffuseGroup selects which)
__device__ __noinline__ f3 shadeClosestHit(const Scene& sc, const FrameParams& fp, const Hit& h, f3 rayDir, bool diffuseGroup) {
  const Tri3 v = getVertices(sc, h.inst, h.prim);
  const Attrib a = interpAttrib(v, h.b1, h.b2);
  const M4 wit = cbLoad3x3(h.inst ? fp.g.WorldIT1 : fp.g.WorldITs0);
  const f3 N = normalize3(mulDir(a.Nrm, wit));
  const f2 rm = getRoughMetal(fp.mat, h.inst, a.UV);
  f3 color = mk3(fp.mat.BaseColors[h.inst][0], fp.mat.BaseColors[h.inst][1], fp.mat.BaseColors[h.inst][2]);
  const f3 V = -rayDir;
  if (rm.y > 0.5f) return reflectionDepth1(sc, rm, N, V, color);
  if (diffuseGroup) color = color * (1.0f - rm.y);
  return diffuseDepth1(sc, N, color);
}

// XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so give each XCD a
// contiguous run of tiles.  A permutation of [0, n): speed only, never correctness.
RT_DEV uint32_t xcdRemap(uint32_t b, uint32_t n) {
  const uint32_t per = n / 8u, full = per * 8u;
  if (b >= full) return b;
  return (b & 7u) * per + (b >> 3);
}

__global__ void __launch_bounds__(256) rayGenKernel(const FrameParams* __restrict__ fpp, const Scene* __restrict__ scp, const unsigned long long* __restrict__ visDepth,
                                                    uint32_t* __restrict__ normalOut, uint16_t* __restrict__ roughMetalOut, uint32_t* __restrict__ velocityOut,
                                                    uint32_t* __restrict__ reflOut, uint32_t* __restrict__ diffOut, unsigned long long* __restrict__ rayCounters,
                                                    uint32_t tilesX, uint32_t numTiles, uint32_t rowBegin, uint32_t rowEnd) {
  const FrameParams& fp = *fpp;
  const Scene& sc = *scp;
  __shared__ int32_t stackMem[RT_STACK * 256];
  __shared__ uint32_t blockRays;
  if (threadIdx.x == 0) blockRays = 0;
  __syncthreads();
  // 16x16 pixel tile per workgroup, 8x8 per wave
  const uint32_t tile = xcdRemap(blockIdx.x, numTiles);
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const uint32_t px = (tile % tilesX) * 16 + (wave & 1u) * 8 + (lane & 7u);
  const uint32_t py = rowBegin + (tile / tilesX) * 16 + (wave >> 1) * 8 + (lane >> 3);
  uint32_t rays = 0;
  if (px < fp.W && py < rowEnd) {
    const uint32_t W = fp.W, H = fp.H;
    const size_t pix = (size_t)py * W + px;
    int32_t* stack = stackMem + threadIdx.x;
    // getPrimarySurface :277-333
    uint32_t visibility = (uint32_t)visDepth[pix];
    f2 screenPos; screenPos.x = ((float)px + 0.5f) / (float)W * 2.0f - 1.0f; screenPos.y = ((float)py + 0.5f) / (float)H * 2.0f - 1.0f;
    screenPos.y = -screenPos.y;
    const f3 eye = mk3(fp.rg.EyePt[0], fp.rg.EyePt[1], fp.rg.EyePt[2]);
    bool hit; f3 N, V, P, color; f2 rghMtl, velocity; uint32_t inst = 0xFFFFFFFFu, prim = 0xFFFFFFFFu;
    if (visibility > 0) {
      --visibility;
      hit = true; inst = visibility >> 24; prim = visibility & 0xFFFFFFu;
      const Tri3 v = getVertices(sc, inst, prim);
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
    normalOut[pix] = packR10G10B10A2(N.x * 0.5f + 0.5f, N.y * 0.5f + 0.5f, N.z * 0.5f + 0.5f, hit ? 1.0f : 0.0f);
    if (hit) roughMetalOut[pix] = (uint16_t)packR8G8(rghMtl.x, rghMtl.y);
    velocityOut[pix] = packR16G16F(velocity.x, velocity.y);

    // getSampleParam :394-406
    uint32_t s = py * W + px;
    s = rng(s); s += fp.g.FrameIndex; s = rng(s); s %= 256u;
    const float xiY = (float)(rng(s) & 0xffffu) / 65536.0f;
    const float cosPhi = sc.cosSin[s], sinPhi = sc.cosSin[256 + s];

    // computeReflection depth 0 :424-484
    f3 refl;
    if (!hit) refl = environment(sc, -V, 0.0f);
    else {
      const float a = rghMtl.x * rghMtl.x;
      const float cosTheta = sqrtf((1.0f - xiY) / (1.0f + (a * a - 1.0f) * xiY));
      const float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
      const f3 Hh = localToWorld(N, mk3(cosPhi * sinTheta, sinPhi * sinTheta, cosTheta));
      const f3 R = reflect3(-V, Hh);
      const float NoL = dot3(N, R);
      if (NoL <= 0.0f) refl = mk3(0.0f, 0.0f, 0.0f);
      else {
        ++rays;
        f3 col = color * rghMtl.y;
        Hit h;
        traceClosest(sc, fp.invWorld[0], fp.invWorld[1], P, R, 1e-5f, 10000.0f, inst, prim, stack, 256, h);
        if (!h.valid) col = environment(sc, R, 0.0f);
        else if (!(col.x <= 0.0f && col.y <= 0.0f && col.z <= 0.0f)) col = shadeClosestHit(sc, fp, h, R, false);
        const f3 f0 = mk3(lerpf(0.04f, color.x, rghMtl.y), lerpf(0.04f, color.y, rghMtl.y), lerpf(0.04f, color.z, rghMtl.y));
        const float NoV = saturatef(dot3(N, V));
        const float VoH = saturatef(dot3(V, Hh));
        const f3 F = fSchlick(f0, VoH);
        const float vis = visSmith(rghMtl.x, NoV, NoL);
        const float NoH = saturatef(dot3(N, Hh));
        const float k = 4.0f * VoH / NoH;
        refl = mk3(col.x * (((NoL * F.x) * vis) * k), col.y * (((NoL * F.y) * vis) * k), col.z * (((NoL * F.z) * vis) * k));
      }
    }
    reflOut[pix] = packR11G11B10F(refl);

    if (rghMtl.y < 1.0f) {   // :559-564
      f3 diff;
      if (!hit) diff = refl;   // same degenerate ray: environment(-V, 0)
      else {
        const float cosTheta = 1.0f - 2.0f * xiY;
        const float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
        const f3 dir = normalize3(N + mk3(cosPhi * sinTheta, sinPhi * sinTheta, cosTheta));
        ++rays;
        f3 col;
        Hit h;
        traceClosest(sc, fp.invWorld[0], fp.invWorld[1], P, dir, 1e-5f, 10000.0f, inst, prim, stack, 256, h);
        if (!h.valid) col = environment(sc, dir, 0.0f);
        else col = shadeClosestHit(sc, fp, h, dir, true);
        diff = col * (color * (1.0f - 0.04f));
      }
      diffOut[pix] = packR11G11B10F(diff);
    }
  }
  if (py < fp.rowBegin || py >= fp.rowEnd) rays = 0;   // apron rows are recomputed by the neighbouring strip's owner: counted there
  if (rays) atomicAdd(&blockRays, rays);
  __syncthreads();
  if (threadIdx.x == 0 && blockRays) {
    atomicAdd(&rayCounters[blockIdx.x & 255u], (unsigned long long)blockRays);           // this frame
    atomicAdd(&rayCounters[256u + (blockIdx.x & 255u)], (unsigned long long)blockRays);    // running total (rtggx_ray_total)
  }
}

int launchRayTrace(rtggx_context* c, const FrameParams& fp, hipStream_t s) {
  uint32_t rb, re;
  passRows(fp, ROWS_GBUFFER, rb, re);
  if (re <= rb) return 0;
  const uint32_t tilesX = (fp.W + 15) / 16, tilesY = (re - rb + 15) / 16;
  const uint32_t numTiles = tilesX * tilesY;
  RT_HIP(hipMemsetAsync(c->rayCounter, 0, 256 * sizeof(unsigned long long), s));
  if (c->timing) hipEventRecord(c->tev[11], s);
  const bool ring = c->kernelRing && c->kevCount < c->kevBegin.size();
  if (ring) hipEventRecord(c->kevBegin[c->kevCount], s);
  hipLaunchKernelGGL(rayGenKernel, dim3(numTiles), dim3(256), 0, s, c->dParams + c->slot, c->dScene, c->visDepth, c->normal, c->roughMetal, c->velocity,
                     c->rtRefl, c->rtDiff, c->rayCounter, tilesX, numTiles, rb, re);
  if (c->timing) hipEventRecord(c->tev[12], s);
  if (ring) hipEventRecord(c->kevEnd[c->kevCount++], s);
  RT_HIP(hipGetLastError());
  return 0;
}

// Test entry: closest-hit queries for an explicit ray list.
__global__ void __launch_bounds__(256) traceRaysKernel(const FrameParams* __restrict__ fpp, const Scene* __restrict__ scp, const float* __restrict__ rays, uint32_t n, float* __restrict__ out) {
  const FrameParams& fp = *fpp;
  const Scene& sc = *scp;
  __shared__ int32_t stackMem[RT_STACK * 256];
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* r = rays + 8 * (size_t)i;
  Hit h;
  traceClosest(sc, fp.invWorld[0], fp.invWorld[1], mk3(r[0], r[1], r[2]), mk3(r[3], r[4], r[5]), r[6], r[7], 0xFFFFFFFFu, 0xFFFFFFFFu,
               stackMem + threadIdx.x, 256, h);
  float* o = out + 6 * (size_t)i;
  o[0] = h.t; o[1] = u2f(h.inst); o[2] = u2f(h.prim); o[3] = h.b1; o[4] = h.b2; o[5] = h.valid ? 1.0f : 0.0f;
}
int launchTraceRays(rtggx_context* c, const FrameParams& fp, const float* dRays, uint32_t n, float* dOut, hipStream_t s) {
  if (!n) return 0;
  hipLaunchKernelGGL(traceRaysKernel, dim3((n + 255) / 256), dim3(256), 0, s, c->dParams + c->slot, c->dScene, dRays, n, dOut);
  RT_HIP(hipGetLastError());
  return 0;
}

}  // namespace rt
