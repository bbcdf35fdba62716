// ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked, imported or called by the product path.
// Scene state, constant-buffer byte layouts and the host-side frame update of the reference:
//   RayTracer::UpdateFrame                      RayTracedGGX/Content/RayTracer.cpp:250-305
//   CBGlobal / RayGenConstants / CBPerObject / CBMaterial        RayTracer.cpp:20-47
//   materials                                   RayTracer.cpp:134-139
//   ground cube                                 RayTracer.cpp:430-496
//   XUSG::IncrementalHalton (closed DLL; algorithm + known answers in SURVEY.md row H4, App. F)
#pragma once
#include <vector>
#include <cstdint>
#include "orc_math.h"
#include "orc_dds.h"

namespace orc {

// ---- constant buffers, byte-for-byte -----------------------------------
struct CBGlobal {                     // 448 B
  float WorldViewProjs[2][16];        // XMStoreFloat4x4(Transpose(world*viewProj))  -> M[i][j] = f[j*4+i]
  float WorldViewProjsPrev[2][16];
  float Worlds[2][12];                // XMStoreFloat3x4(world)                       -> M[i][j] = f[j*4+i], j<3
  float WorldITs0[12];                // XMStoreFloat3x4(I)
  float WorldIT1[11];                 // XMStoreFloat3x4(rot), 12th float lands on FrameIndex and is overwritten
  uint32_t FrameIndex;
};
struct RayGenConstants { float ProjToWorld[16]; float EyePt[4]; float ProjBias[2]; float pad[2]; };   // 96 B
struct CBPerObject { float WorldViewProj[16]; float ProjBias[2]; float pad[2]; };                        // 80 B
struct CBMaterial { float BaseColors[2][4]; float RoughMetals[2][4]; };                                   // 64 B
struct FrameConstants { CBGlobal g; RayGenConstants rg; CBPerObject po[2]; CBMaterial mat; };             // 768 B
static_assert(sizeof(CBGlobal) == 448 && sizeof(FrameConstants) == 768, "constant-buffer layout");

static inline M4 cb_load4x4(const float* f) { M4 r; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = f[j * 4 + i]; return r; }
static inline M4 cb_load4x3(const float* f) {   // float4x3 (Worlds): 3 columns of 4
  M4 r = identity();
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = f[j * 4 + i];
  r.m[0][3] = r.m[1][3] = r.m[2][3] = 0.0f; r.m[3][3] = 1.0f;
  return r;
}
static inline M4 cb_load3x3(const float* f) {   // float3x3 (WorldITs): 3 padded columns of 3
  M4 r = identity();
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = f[j * 4 + i];
  return r;
}
static inline void cb_store4x4T(float* f, const M4& m) { for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) f[j * 4 + i] = m.m[i][j]; }
static inline void cb_store3x4(float* f, const M4& m, int count = 12) {   // XMStoreFloat3x4: rows of the transpose
  for (int k = 0; k < count; ++k) { const int r = k / 4, c = k % 4; f[k] = m.m[c][r]; }
}

// ---- BVH (format shared with the product so the same arrays can be re-traced on the CPU) -------
struct BvhNode {           // 64 B; internal nodes only, node 0 is the root
  float lmin[3], lmax[3];  // box of the left child
  float rmin[3], rmax[3];  // box of the right child
  int32_t left, right;     // >= 0: internal node index; < 0: leaf, ~ref = slot in the triangle array
  int32_t pad[2];
};
struct BvhTri {            // 64 B, leaf order
  float v0[3], v1[3], v2[3];
  uint32_t pad0[3];
  uint32_t prim;           // word 12: primitive index in the index buffer (= SV_PrimitiveID / PrimitiveIndex())
  uint32_t pad1[3];
};
static_assert(sizeof(BvhNode) == 64 && sizeof(BvhTri) == 64, "bvh layout");
struct Bvh { std::vector<BvhNode> nodes; std::vector<BvhTri> tris; int32_t root = -1; };

struct Mesh { std::vector<float> verts; std::vector<uint32_t> idx; Bvh bvh; };

enum BufferId {
  BUF_VISIBILITY = 0, BUF_DEPTH, BUF_NORMAL, BUF_ROUGH_METAL, BUF_VELOCITY, BUF_RT_REFL, BUF_RT_DIFF,
  BUF_TSS0, BUF_TSS1, BUF_FLT_RFL, BUF_FLT_DFF, BUF_BACKBUFFER, BUF_SH_COEFFS, BUF_COUNT
};

struct Ctx {
  uint32_t W = 0, H = 0;
  Mesh mesh[2];
  EnvMap env;
  float sh[9][3] = {};
  float cosTab[256], sinTab[256];          // cos/sin(2*pi*s/256), double libm rounded to fp32
  FrameConstants fc{};
  M4 invWorld[2];                          // TLAS instance data: world -> object, from fc.g.Worlds
  // render targets
  std::vector<uint32_t> vis, depth, normal, velocity, refl, diff, backbuffer;
  std::vector<uint16_t> roughMetal;
  std::vector<uint64_t> tss[2], fltRfl, fltDff;
  uint32_t frameParity = 0;                // Denoiser::m_frameParity (Denoiser.cpp:12,69)
  uint64_t rayCount = 0;                   // non-degenerate TraceRay calls of the last RayTrace
  // host state of UpdateFrame
  float posScale[4] = {0, 0, 0, 1};
  uint32_t haltonBase2 = 0, haltonBase3 = 0; float haltonX = 0, haltonY = 0;
  float angle = 0.0f; uint32_t sFrameIndex = 0; bool havePrev = false;
  float prevWVP[2][16];
  int threads = 1;
  bool vndf = false;                       // the product's opt-in sampler (orc_raytrace.h vndf_half_vector)
};

// XUSG::IncrementalHalton: n-th call returns (radical_inverse_2(n), radical_inverse_3(n)),
// accumulated incrementally in fp32.
static inline void incremental_halton(Ctx& c, float& hx, float& hy) {
  {  // base 2: flip bits from the least significant one
    uint32_t oldBase = c.haltonBase2++;
    uint32_t diff = c.haltonBase2 ^ oldBase;
    float s = 0.5f;
    do {
      if (oldBase & 1u) c.haltonX -= s; else c.haltonX += s;
      s *= 0.5f; diff >>= 1; oldBase >>= 1;
    } while (diff);
  }
  {  // base 3: two bits per digit
    uint32_t mask = 0x3u, add = 0x1u;
    float s = 1.0f / 3.0f;
    ++c.haltonBase3;
    for (;;) {
      if ((c.haltonBase3 & mask) == mask) {
        c.haltonBase3 += add;                 // carry into the next digit
        c.haltonY += -2.0f * s;
        mask <<= 2; add <<= 2; s *= 1.0f / 3.0f;
      } else { c.haltonY += s; break; }
    }
  }
  hx = c.haltonX; hy = c.haltonY;
}

static inline void set_default_materials(Ctx& c) {   // RayTracer.cpp:134-139
  const float bc0[4] = {0.95f, 0.93f, 0.88f, 1.0f}, bc1[4] = {1.0f, 0.71f, 0.29f, 1.0f};
  const float rm0[4] = {0.5f, 1.0f, 0.0f, 0.0f}, rm1[4] = {0.16f, 1.0f, 0.0f, 0.0f};
  std::memcpy(c.fc.mat.BaseColors[0], bc0, 16); std::memcpy(c.fc.mat.BaseColors[1], bc1, 16);
  std::memcpy(c.fc.mat.RoughMetals[0], rm0, 16); std::memcpy(c.fc.mat.RoughMetals[1], rm1, 16);
}

static inline void set_ground_mesh(Mesh& m) {   // RayTracer.cpp:430-496
  static const float v[24][6] = {
    {-1, 1, -1, 0, 1, 0}, {1, 1, -1, 0, 1, 0}, {1, 1, 1, 0, 1, 0}, {-1, 1, 1, 0, 1, 0},
    {-1, -1, -1, 0, -1, 0}, {1, -1, -1, 0, -1, 0}, {1, -1, 1, 0, -1, 0}, {-1, -1, 1, 0, -1, 0},
    {-1, -1, 1, -1, 0, 0}, {-1, -1, -1, -1, 0, 0}, {-1, 1, -1, -1, 0, 0}, {-1, 1, 1, -1, 0, 0},
    {1, -1, 1, 1, 0, 0}, {1, -1, -1, 1, 0, 0}, {1, 1, -1, 1, 0, 0}, {1, 1, 1, 1, 0, 0},
    {-1, -1, -1, 0, 0, -1}, {1, -1, -1, 0, 0, -1}, {1, 1, -1, 0, 0, -1}, {-1, 1, -1, 0, 0, -1},
    {-1, -1, 1, 0, 0, 1}, {1, -1, 1, 0, 0, 1}, {1, 1, 1, 0, 0, 1}, {-1, 1, 1, 0, 0, 1}};
  static const uint32_t idx[36] = {3, 1, 0, 2, 1, 3, 6, 4, 5, 7, 4, 6, 11, 9, 8, 10, 9, 11,
                                   14, 12, 13, 15, 12, 14, 19, 17, 16, 18, 17, 19, 22, 20, 21, 23, 20, 22};
  m.verts.assign(&v[0][0], &v[0][0] + 144);
  m.idx.assign(idx, idx + 36);
}

// RayTracer::UpdateFrame (RayTracer.cpp:250-305).  eye = float3, viewProj = view*proj (row-major M4).
static inline void update_frame(Ctx& c, const float eye[3], const M4& viewProj, float timeStep) {
  float hx, hy; incremental_halton(c, hx, hy);
  const float bias[2] = {(hx * 2.0f - 1.0f) / (float)c.W, (hy * 2.0f - 1.0f) / (float)c.H};      // :254-258
  {                                                                                              // :260-267
    const M4 projToWorld = inverse(viewProj);
    cb_store4x4T(c.fc.rg.ProjToWorld, projToWorld);
    c.fc.rg.EyePt[0] = eye[0]; c.fc.rg.EyePt[1] = eye[1]; c.fc.rg.EyePt[2] = eye[2]; c.fc.rg.EyePt[3] = 0.0f;
    c.fc.rg.ProjBias[0] = bias[0]; c.fc.rg.ProjBias[1] = bias[1]; c.fc.rg.pad[0] = c.fc.rg.pad[1] = 0.0f;
  }
  c.angle += 16.0f * timeStep * 3.141592654f / 180.0f;                                           // :270-271
  const M4 rot = rotation_y(c.angle);
  const M4 worlds[2] = {                                                                         // :274-279
    mul(scaling(10.0f, 0.5f, 10.0f), translation(0.0f, -0.5f, 0.0f)),
    mul(mul(scaling(c.posScale[3], c.posScale[3], c.posScale[3]), rot), translation(c.posScale[0], c.posScale[1], c.posScale[2]))};
  for (int i = 0; i < 2; ++i) {                                                                  // :285-293
    float wvp[16]; cb_store4x4T(wvp, mul(worlds[i], viewProj));
    // m_worldViewProjs is uninitialised on the first frame in the reference; defined here as prev = current
    std::memcpy(c.fc.g.WorldViewProjsPrev[i], c.havePrev ? c.prevWVP[i] : wvp, 64);
    std::memcpy(c.fc.g.WorldViewProjs[i], wvp, 64);
    cb_store3x4(c.fc.g.Worlds[i], worlds[i]);
    std::memcpy(c.prevWVP[i], wvp, 64);
    std::memcpy(c.fc.po[i].WorldViewProj, wvp, 64);                                              // :298-303
    c.fc.po[i].ProjBias[0] = bias[0]; c.fc.po[i].ProjBias[1] = bias[1]; c.fc.po[i].pad[0] = c.fc.po[i].pad[1] = 0.0f;
  }
  cb_store3x4(c.fc.g.WorldITs0, identity());
  cb_store3x4(c.fc.g.WorldIT1, rot, 11);
  c.havePrev = true;
  c.fc.g.FrameIndex = c.sFrameIndex++;                                                           // :294-295
  c.sFrameIndex %= 256u;
}

// RayTracer::UpdateAccelerationStructure (RayTracer.cpp:326-341): refresh the two instance
// transforms of the TLAS.  The software TLAS keeps world->object matrices.
static inline void update_as(Ctx& c) {
  for (int i = 0; i < 2; ++i) c.invWorld[i] = inverse(cb_load4x3(c.fc.g.Worlds[i]));
}

}  // namespace orc
