// ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked, imported or called by the product path.
// Restatement of the ray-tracing library RayTracedGGX/Content/Shaders/RayTracing.hlsl
// (raygenMain :541-565, closestHitReflection :571-590, closestHitDiffuse :593-614, missMain
// :620-625 and every helper they call), Material.hlsli, BRDFModels.hlsli and
// XUSG/Shaders/SHIrradianceTypeless.hlsli:16-37.  min16float is evaluated as fp32
// (SURVEY.md App. A).  D3D-defined pieces restated from the D3D functional spec: cube-map face
// selection and (s,t) mapping, bilinear + mip-linear filtering with full fp32 weights, seamless
// cube edges (out-of-face taps are re-projected onto the neighbouring face, nearest texel).
// Choices: cos/sin(2*pi*xi.x) come from a 256-entry table (xi.x has 256 values);
// pow(1-VoH, 5) is evaluated by multiplication.
#pragma once
#include "orc_scene.h"
#include "orc_bvh.h"
#include "orc_formats.h"

namespace orc {

static const float kPI = 3.1415926535897f;   // BRDFModels.hlsli:5

// ---- environment map (RayTracing.hlsl:167-180, sampler ANISOTROPIC_WRAP, RayTracer.cpp:529) ----
static inline void cube_face_uv(float3 d, int& face, float& u, float& v) {
  const float ax = std::fabs(d.x), ay = std::fabs(d.y), az = std::fabs(d.z);
  if (ax >= ay && ax >= az) { face = d.x >= 0.0f ? 0 : 1; u = (d.x >= 0.0f ? -d.z : d.z) / ax; v = -d.y / ax; }
  else if (ay >= az) { face = d.y >= 0.0f ? 2 : 3; u = d.x / ay; v = (d.y >= 0.0f ? d.z : -d.z) / ay; }
  else { face = d.z >= 0.0f ? 4 : 5; u = (d.z >= 0.0f ? d.x : -d.x) / az; v = -d.y / az; }
}
static inline float3 cube_face_dir(int face, float u, float v) {
  switch (face) {
    case 0: return f3(1.0f, -v, -u);
    case 1: return f3(-1.0f, -v, u);
    case 2: return f3(u, 1.0f, v);
    case 3: return f3(u, -1.0f, -v);
    case 4: return f3(u, -v, 1.0f);
    default: return f3(-u, -v, -1.0f);
  }
}
static inline float3 cube_texel(const EnvMap& e, uint32_t mip, int face, int x, int y) {
  const int s = (int)(e.size >> mip ? e.size >> mip : 1);
  if (x < 0 || y < 0 || x >= s || y >= s) {
    // seamless edge: re-project the texel centre through the cube onto the adjacent face
    const float u = ((float)x + 0.5f) / (float)s * 2.0f - 1.0f, v = ((float)y + 0.5f) / (float)s * 2.0f - 1.0f;
    float uu, vv; cube_face_uv(cube_face_dir(face, u, v), face, uu, vv);
    x = (int)std::floor((uu * 0.5f + 0.5f) * (float)s); y = (int)std::floor((vv * 0.5f + 0.5f) * (float)s);
    x = std::min(std::max(x, 0), s - 1); y = std::min(std::max(y, 0), s - 1);
  }
  const uint16_t* t = &e.level[(size_t)mip * 6 + (size_t)face][4 * ((size_t)y * s + x)];
  return f3(f16_to_f32(t[0]), f16_to_f32(t[1]), f16_to_f32(t[2]));
}
static inline float3 cube_bilinear(const EnvMap& e, uint32_t mip, int face, float u, float v) {
  const int s = (int)(e.size >> mip ? e.size >> mip : 1);
  const float x = (u * 0.5f + 0.5f) * (float)s - 0.5f, y = (v * 0.5f + 0.5f) * (float)s - 0.5f;
  const float x0 = std::floor(x), y0 = std::floor(y);
  const float fx = x - x0, fy = y - y0;
  const int ix = (int)x0, iy = (int)y0;
  const float w00 = (1.0f - fx) * (1.0f - fy), w10 = fx * (1.0f - fy), w01 = (1.0f - fx) * fy, w11 = fx * fy;
  const float3 c00 = cube_texel(e, mip, face, ix, iy), c10 = cube_texel(e, mip, face, ix + 1, iy);
  const float3 c01 = cube_texel(e, mip, face, ix, iy + 1), c11 = cube_texel(e, mip, face, ix + 1, iy + 1);
  return ((c00 * w00 + c10 * w10) + c01 * w01) + c11 * w11;
}
static inline float3 environment(const Ctx& c, float3 dir, float level) {
  int face; float u, v; cube_face_uv(dir, face, u, v);
  const float maxLevel = (float)(c.env.mips - 1);
  const float l = std::fmin(std::fmax(level, 0.0f), maxLevel);
  const float l0 = std::floor(l), fl = l - l0;
  const uint32_t m0 = (uint32_t)l0, m1 = std::min(m0 + 1, c.env.mips - 1);
  const float3 a = cube_bilinear(c.env, m0, face, u, v);
  if (fl == 0.0f) return a;
  const float3 b = cube_bilinear(c.env, m1, face, u, v);
  return a * (1.0f - fl) + b * fl;
}

// ---- SH (SHIrradianceTypeless.hlsli:16-37) ------------------------------------------------------
static inline float3 evaluate_sh_irradiance(const Ctx& c, float3 norm) {
  const float c1 = 0.42904276540489171563379376569857f, c2 = 0.51166335397324424423977581244463f;
  const float c3 = 0.24770795610037568833406429782001f, c4 = 0.88622692545275801364908374167057f;
  const float x = -norm.x, y = -norm.y, z = norm.z;
  auto L = [&](int i) { return f3(c.sh[i][0], c.sh[i][1], c.sh[i][2]); };
  float3 irr = (c1 * (x * x - y * y)) * L(8);
  irr = irr + (c3 * (3.0f * z * z - 1.0f)) * L(6);
  irr = irr + c4 * L(0);
  irr = irr + (2.0f * c1) * ((L(4) * x * y + L(7) * x * z) + L(5) * y * z);
  irr = irr + (2.0f * c2) * ((L(3) * x + L(1) * y) + L(2) * z);
  return f3(std::fmax(0.0f, irr.x), std::fmax(0.0f, irr.y), std::fmax(0.0f, irr.z));
}
// SphericalHarmonics::Transform (closed XUSG.dll; "parity unpinned").  Projects mip 0 of the cube
// onto the 9 real orthonormal SH basis functions in the consumer's axis convention
// (x,y,z) = (-d.x,-d.y,d.z), texel solid-angle weights normalised to 4*pi, accumulated in double.
static inline void transform_sh(Ctx& c) {
  double acc[9][3] = {}; double wsum = 0.0;
  const int s = (int)c.env.size;
  for (int face = 0; face < 6; ++face) for (int y = 0; y < s; ++y) for (int x = 0; x < s; ++x) {
    const double u = ((double)x + 0.5) / s * 2.0 - 1.0, v = ((double)y + 0.5) / s * 2.0 - 1.0;
    const float3 df = cube_face_dir(face, (float)u, (float)v);
    const double len = std::sqrt((double)df.x * df.x + (double)df.y * df.y + (double)df.z * df.z);
    const double dx = -df.x / len, dy = -df.y / len, dz = df.z / len;
    const double w = 1.0 / (len * len * len);
    const double Y[9] = {0.28209479177387814, 0.4886025119029199 * dy, 0.4886025119029199 * dz, 0.4886025119029199 * dx,
                         1.0925484305920792 * dx * dy, 1.0925484305920792 * dy * dz, 0.31539156525252005 * (3.0 * dz * dz - 1.0),
                         1.0925484305920792 * dx * dz, 0.5462742152960396 * (dx * dx - dy * dy)};
    const float3 L = cube_texel(c.env, 0, face, x, y);
    for (int i = 0; i < 9; ++i) { acc[i][0] += Y[i] * w * L.x; acc[i][1] += Y[i] * w * L.y; acc[i][2] += Y[i] * w * L.z; }
    wsum += w;
  }
  const double norm = 4.0 * 3.14159265358979323846 / wsum;
  for (int i = 0; i < 9; ++i) for (int k = 0; k < 3; ++k) c.sh[i][k] = (float)(acc[i][k] * norm);
}

// ---- Material.hlsli ----------------------------------------------------------------------------
static inline float2 get_uv(float3 n, float3 p, float3 scl) {   // :16-23
  float2 uv = {std::fabs(n.x) * p.y * scl.y, std::fabs(n.x) * p.z * scl.z};
  uv.x += std::fabs(n.y) * p.z * scl.z; uv.y += std::fabs(n.y) * p.x * scl.x;
  uv.x += std::fabs(n.z) * p.x * scl.x; uv.y += std::fabs(n.z) * p.y * scl.y;
  return {uv.x * 0.5f + 0.5f, uv.y * 0.5f + 0.5f};
}
static inline float2 get_rough_metal(const Ctx& c, uint32_t inst, float2 uv) {   // :30-48
  float rough = c.fc.mat.RoughMetals[inst][0];
  if (inst == 0) {
    const uint32_t px = ftou(uv.x * 5.0f) & 1u, py = ftou(uv.y * 5.0f) & 1u;
    rough = (px ^ py) ? rough * 0.25f : rough;
  }
  return {rough, c.fc.mat.RoughMetals[inst][1]};
}

// ---- BRDFModels.hlsli --------------------------------------------------------------------------
static inline float vis_smith(float roughness, float NoV, float NoL) {   // :30-39
  const float a = roughness * roughness, a2 = a * a;
  const float v = NoV + std::sqrt(NoV * (NoV - NoV * a2) + a2);
  const float l = NoL + std::sqrt(NoL * (NoL - NoL * a2) + a2);
  return 1.0f / (v * l);
}
static inline float3 f_schlick(float3 spec, float VoH) {   // :54-62
  const float x = 1.0f - VoH, x2 = x * x;
  const float fc = (x2 * x2) * x;
  const float s = saturate(50.0f * spec.y) * fc;
  return f3(s + (1.0f - fc) * spec.x, s + (1.0f - fc) * spec.y, s + (1.0f - fc) * spec.z);
}
static inline float3 env_brdf_approx(float3 spec, float roughness, float NoV) {   // :64-77
  const float rx = roughness * -1.0f + 1.0f, ry = roughness * -0.0275f + 0.0425f;
  const float rz = roughness * -0.572f + 1.04f, rw = roughness * 0.022f + -0.04f;
  const float a004 = std::fmin(rx * rx, exp2Contract(-9.28f * NoV)) * rx + ry;
  const float ABx = -1.04f * a004 + rz;
  float ABy = 1.04f * a004 + rw;
  ABy *= saturate(50.0f * spec.y);
  return f3(spec.x * ABx + ABy, spec.y * ABx + ABy, spec.z * ABx + ABy);
}

// ---- RayTracing.hlsl ---------------------------------------------------------------------------
static inline uint32_t rng(uint32_t seed) {   // :379-387
  seed = seed * 747796405u + 1u;
  seed = ((seed >> ((seed >> 28) + 4u)) ^ seed) * 277803737u;
  seed = (seed >> 22) ^ seed;
  return seed;
}
struct SampleParam { uint32_t s; float x, y; };
static inline SampleParam get_sample_param(uint32_t px, uint32_t py, uint32_t W, uint32_t frameIndex) {   // :394-406
  uint32_t s = py * W + px;
  s = rng(s); s += frameIndex; s = rng(s); s %= 256u;
  return {s, (float)s / 256.0f, (float)(rng(s) & 0xffffu) / 65536.0f};
}
static inline float calc_mip_from_roughness(float rgh, float mipCount) {   // :416-422
  const float level = 3.0f - 1.15f * log2Contract(rgh);
  return mipCount - 1.0f - level;
}
static inline float3 local_to_world(float3 n, float3 l) {   // computeLocalToWorld + combine :129-147
  const float3 up = std::fabs(n.y) < 0.999f ? f3(0, 1, 0) : f3(1, 0, 0);
  const float3 xAxis = normalize(cross(up, n));
  const float3 yAxis = cross(n, xAxis);
  return (xAxis * l.x + yAxis * l.y) + n * l.z;
}

// Opt-in sampler of the product (rtggx_set_sampler), restated: the half vector from the distribution of visible normals (Heitz 2018,
// "Sampling the GGX Distribution of Visible Normals", listing 1; isotropic, alpha = roughness^2), in the tangent frame of local_to_world.
// Not the reference's sampler (that is computeLocalDirectionGGX, the default): north_star names it, SURVEY R4 scopes it as an opt-in variant.
static inline float3 vndf_half_vector(float3 n, float3 v, float alpha, float cosPhi, float sinPhi, float u) {
  const float3 up = std::fabs(n.y) < 0.999f ? f3(0, 1, 0) : f3(1, 0, 0);
  const float3 xAxis = normalize(cross(up, n));
  const float3 yAxis = cross(n, xAxis);
  const float3 ve = f3(dot(v, xAxis), dot(v, yAxis), dot(v, n));
  const float3 vh = normalize(f3(alpha * ve.x, alpha * ve.y, ve.z));
  const float lensq = vh.x * vh.x + vh.y * vh.y;
  const float3 t1v = lensq > 0.0f ? f3(-vh.y, vh.x, 0.0f) * (1.0f / std::sqrt(lensq)) : f3(1, 0, 0);
  const float3 t2v = cross(vh, t1v);
  const float r = std::sqrt(u);
  const float t1 = r * cosPhi;
  float t2 = r * sinPhi;
  const float sw = 0.5f * (1.0f + vh.z);
  t2 = (1.0f - sw) * std::sqrt(1.0f - t1 * t1) + sw * t2;
  const float3 nh = (t1 * t1v + t2 * t2v) + std::sqrt(std::fmax(0.0f, (1.0f - t1 * t1) - t2 * t2)) * vh;
  const float3 hl = normalize(f3(alpha * nh.x, alpha * nh.y, std::fmax(0.0f, nh.z)));
  return (xAxis * hl.x + yAxis * hl.y) + n * hl.z;
}
struct Vertex3 { float3 pos[3], nrm[3]; };
static inline Vertex3 get_vertices(const Ctx& c, uint32_t inst, uint32_t prim) {   // :230-244
  Vertex3 v; const Mesh& m = c.mesh[inst];
  for (int k = 0; k < 3; ++k) {
    const float* p = &m.verts[6 * (size_t)m.idx[3 * (size_t)prim + k]];
    v.pos[k] = f3(p[0], p[1], p[2]); v.nrm[k] = f3(p[3], p[4], p[5]);
  }
  return v;
}
struct Attrib { float3 Pos, Nrm; float2 UV; };
static inline Attrib interp_attrib(const Vertex3& v, float b1, float b2) {   // :249-271
  const float w0 = 1.0f - (b1 + b2);
  Attrib a;
  a.Pos = (w0 * v.pos[0] + b1 * v.pos[1]) + b2 * v.pos[2];
  a.Nrm = (w0 * v.nrm[0] + b1 * v.nrm[1]) + b2 * v.nrm[2];
  a.UV = get_uv(a.Nrm, a.Pos, f3(1.0f, 0.2f, 1.0f));
  return a;
}
static inline float2 calc_barycentrics(const float4 p[3], float2 ndc) {   // :204-225
  const float3 invW = f3(1.0f / p[0].w, 1.0f / p[1].w, 1.0f / p[2].w);
  const float2 ndc0 = {p[0].x * invW.x, p[0].y * invW.x}, ndc1 = {p[1].x * invW.y, p[1].y * invW.y}, ndc2 = {p[2].x * invW.z, p[2].y * invW.z};
  const float det = (ndc2.x - ndc1.x) * (ndc0.y - ndc1.y) - (ndc2.y - ndc1.y) * (ndc0.x - ndc1.x);
  const float invDet = 1.0f / det;
  const float3 dPdx = f3(ndc1.y - ndc2.y, ndc2.y - ndc0.y, ndc0.y - ndc1.y) * invDet;
  const float3 dPdy = f3(ndc2.x - ndc1.x, ndc0.x - ndc2.x, ndc1.x - ndc0.x) * invDet;
  const float2 dv = {ndc.x - ndc0.x, ndc.y - ndc0.y};
  const float interpInvW = (invW.x + dv.x * dot(invW, dPdx)) + dv.y * dot(invW, dPdy);
  const float interpW = 1.0f / interpInvW;
  return {interpW * (dv.x * dPdx.y * invW.y + dv.y * dPdy.y * invW.y),
          interpW * (dv.x * dPdx.z * invW.z + dv.y * dPdy.z * invW.z)};
}

struct Surface { bool hit; float3 N, V, P; float3 color; float2 rghMtl; float2 velocity; uint32_t inst, prim; };

// computeReflection at recursion depth 1 (called from the closest-hit shaders) :424-484
static inline float3 reflection_depth1(const Ctx& c, float2 rghMtl, float3 N, float3 V, float3 color) {
  const float level = calc_mip_from_roughness(rghMtl.x, (float)c.env.mips);
  const float a = rghMtl.x * rghMtl.x;
  const float3 R = reflect(-V, N);
  const float3 dir = lerp(N, R, (1.0f - a) * (std::sqrt(1.0f - a) + a));
  const float NoL = dot(N, dir);
  if (NoL <= 0.0f) return f3(0, 0, 0);
  const float3 env = environment(c, dir, level);
  const float3 f0 = f3(lerp(0.04f, color.x, rghMtl.y), lerp(0.04f, color.y, rghMtl.y), lerp(0.04f, color.z, rghMtl.y));
  const float NoV = saturate(dot(N, V));
  return env * env_brdf_approx(f0, rghMtl.x, NoV);
}
// computeDiffuse at recursion depth 1 :486-535
static inline float3 diffuse_depth1(const Ctx& c, float3 N, float3 color) {
  const float3 irr = evaluate_sh_irradiance(c, N);
  return f3(irr.x / kPI, irr.y / kPI, irr.z / kPI) * color;
}
// Surface data of a closest hit (:575-585 / :595-606)
static inline void hit_surface(const Ctx& c, const Hit& h, float3& N, float2& rghMtl, float3& color) {
  const Vertex3 v = get_vertices(c, h.inst, h.prim);
  const Attrib a = interp_attrib(v, h.b1, h.b2);
  const M4 wit = cb_load3x3(h.inst ? c.fc.g.WorldIT1 : c.fc.g.WorldITs0);
  N = normalize(mul_dir(a.Nrm, wit));
  rghMtl = get_rough_metal(c, h.inst, a.UV);
  color = f3(c.fc.mat.BaseColors[h.inst][0], c.fc.mat.BaseColors[h.inst][1], c.fc.mat.BaseColors[h.inst][2]);
}

// One pixel of raygenMain (:541-565).  Returns the number of non-degenerate rays traced.
static inline uint32_t raygen_pixel(Ctx& c, uint32_t px, uint32_t py) {
  const uint32_t W = c.W, H = c.H; const size_t pix = (size_t)py * W + px;
  const FrameConstants& fc = c.fc;
  uint32_t rays = 0;
  // getPrimarySurface :277-333
  Surface s{};
  uint32_t visibility = c.vis[pix];
  float2 screenPos = {((float)px + 0.5f) / (float)W * 2.0f - 1.0f, ((float)py + 0.5f) / (float)H * 2.0f - 1.0f};
  screenPos.y = -screenPos.y;
  const float3 eye = f3(fc.rg.EyePt[0], fc.rg.EyePt[1], fc.rg.EyePt[2]);
  if (visibility > 0) {
    --visibility;
    s.hit = true; s.inst = visibility >> 24; s.prim = visibility & 0xFFFFFFu;
    const Vertex3 v = get_vertices(c, s.inst, s.prim);
    const M4 wvp = cb_load4x4(fc.g.WorldViewProjs[s.inst]);
    float4 p[3];
    for (int k = 0; k < 3; ++k) p[k] = mul_point(v.pos[k], wvp);
    screenPos.x -= fc.rg.ProjBias[0]; screenPos.y -= fc.rg.ProjBias[1];
    const float2 bary = calc_barycentrics(p, screenPos);
    const Attrib a = interp_attrib(v, bary.x, bary.y);
    s.color = f3(fc.mat.BaseColors[s.inst][0], fc.mat.BaseColors[s.inst][1], fc.mat.BaseColors[s.inst][2]);
    s.rghMtl = get_rough_metal(c, s.inst, a.UV);
    const float4 hPrev = mul_point(a.Pos, cb_load4x4(fc.g.WorldViewProjsPrev[s.inst]));
    s.velocity = {(screenPos.x - hPrev.x / hPrev.w) * 0.5f, (screenPos.y - hPrev.y / hPrev.w) * -0.5f};
    const float4 P4 = mul_point(a.Pos, cb_load4x3(fc.g.Worlds[s.inst]));
    s.P = f3(P4.x, P4.y, P4.z);
    s.N = normalize(mul_dir(a.Nrm, cb_load3x3(s.inst ? fc.g.WorldIT1 : fc.g.WorldITs0)));
    s.V = normalize(eye - s.P);
  } else {
    const float4 world = mul_vec4(float4{screenPos.x, screenPos.y, 0.0f, 1.0f}, cb_load4x4(fc.rg.ProjToWorld));
    s.hit = false; s.velocity = {0.0f, 0.0f};
    s.P = f3(world.x / world.w, world.y / world.w, world.z / world.w);
    s.N = f3(0, 0, 0);
    s.V = normalize(eye - s.P);
    s.rghMtl = {0.0f, 0.0f};   // rghMtl.x is left unset by the reference; never consumed on this path
    s.color = f3(0, 0, 0);
  }
  // G-buffer stores :552-554
  c.normal[pix] = pack_r10g10b10a2(s.N.x * 0.5f + 0.5f, s.N.y * 0.5f + 0.5f, s.N.z * 0.5f + 0.5f, s.hit ? 1.0f : 0.0f);
  if (s.hit) c.roughMetal[pix] = pack_r8g8(s.rghMtl.x, s.rghMtl.y);
  c.velocity[pix] = pack_r16g16f(s.velocity.x, s.velocity.y);

  const SampleParam xi = get_sample_param(px, py, W, fc.g.FrameIndex);

  // computeReflection, depth 0 :424-484
  float3 refl;
  if (!s.hit) refl = environment(c, -s.V, 0.0f);          // degenerate ray [0,0] always misses -> missMain
  else {
    const float a = s.rghMtl.x * s.rghMtl.x;
    // computeLocalDirectionGGX :92-101 with cos/sin(2*pi*xi.x) from the table
    float3 Hh;
    if (c.vndf) Hh = vndf_half_vector(s.N, s.V, a, c.cosTab[xi.s], c.sinTab[xi.s], xi.y);
    else {
      const float cosTheta = std::sqrt((1.0f - xi.y) / (1.0f + (a * a - 1.0f) * xi.y));
      const float sinTheta = std::sqrt(1.0f - cosTheta * cosTheta);
      Hh = local_to_world(s.N, f3(c.cosTab[xi.s] * sinTheta, c.sinTab[xi.s] * sinTheta, cosTheta));
    }
    const float3 R = reflect(-s.V, Hh);
    const float NoL = dot(s.N, R);
    if (NoL <= 0.0f) refl = f3(0, 0, 0);                  // :459
    else {
      ++rays;
      float3 col = s.color * s.rghMtl.y;                  // payload preset :456
      const Hit h = trace_closest(c, s.P, R, 1e-5f, 10000.0f, s.inst, s.prim);
      if (!h.valid) col = environment(c, R, 0.0f);        // missMain :620-625
      else if (!(col.x <= 0.0f && col.y <= 0.0f && col.z <= 0.0f)) {   // closestHitReflection :573
        float3 N2, color2; float2 rm2; hit_surface(c, h, N2, rm2, color2);
        const float3 V2 = -R;
        if (rm2.y > 0.5f) col = reflection_depth1(c, rm2, N2, V2, color2);
        else col = diffuse_depth1(c, N2, color2);
      }
      const float3 f0 = f3(lerp(0.04f, s.color.x, s.rghMtl.y), lerp(0.04f, s.color.y, s.rghMtl.y), lerp(0.04f, s.color.z, s.rghMtl.y));
      const float NoV = saturate(dot(s.N, s.V));
      const float VoH = saturate(dot(s.V, Hh));
      const float3 F = f_schlick(f0, VoH);
      const float vis = vis_smith(s.rghMtl.x, NoV, NoL);
      const float NoH = saturate(dot(s.N, Hh));
      const float k = 4.0f * VoH / NoH;
      refl = f3(col.x * (((NoL * F.x) * vis) * k), col.y * (((NoL * F.y) * vis) * k), col.z * (((NoL * F.z) * vis) * k));   // :477
      if (c.vndf) {      // F x G2 / G1(V) = F x G1(L), separable Smith terms of Vis_Smith
        const float a2 = a * a;
        const float g1l = (2.0f * NoL) / (NoL + std::sqrt(NoL * (NoL - NoL * a2) + a2));
        refl = f3(col.x * (F.x * g1l), col.y * (F.y * g1l), col.z * (F.z * g1l));
      }
    }
  }
  c.refl[pix] = pack_r11g11b10f(refl.x, refl.y, refl.z);

  if (s.rghMtl.y < 1.0f) {   // :559-564, computeDiffuse depth 0 :486-535
    float3 diff;
    if (!s.hit) diff = environment(c, -s.V, 0.0f);
    else {
      // computeDirectionCos :150-162 (uniform-sphere branch)
      const float cosTheta = 1.0f - 2.0f * xi.y;
      const float sinTheta = std::sqrt(1.0f - cosTheta * cosTheta);
      const float3 dir = normalize(s.N + f3(c.cosTab[xi.s] * sinTheta, c.sinTab[xi.s] * sinTheta, cosTheta));
      ++rays;
      float3 col;
      const Hit h = trace_closest(c, s.P, dir, 1e-5f, 10000.0f, s.inst, s.prim);
      if (!h.valid) col = environment(c, dir, 0.0f);
      else {                                              // closestHitDiffuse :593-614
        float3 N2, color2; float2 rm2; hit_surface(c, h, N2, rm2, color2);
        const float3 V2 = -dir;
        if (rm2.y > 0.5f) col = reflection_depth1(c, rm2, N2, V2, color2);
        else col = diffuse_depth1(c, N2, color2 * (1.0f - rm2.y));
      }
      diff = col * (s.color * (1.0f - 0.04f));            // :532
    }
    c.diff[pix] = pack_r11g11b10f(diff.x, diff.y, diff.z);
  }
  return rays;
}

}  // namespace orc
