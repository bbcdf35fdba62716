// ORACLE -- TEST INFRASTRUCTURE ONLY.
// C entry points of the CPU oracle (loaded with ctypes from tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg -- and from nowhere else).  The product path (raytracedggx_amd/,
// include/rtggx.h) never includes, links or calls anything in this directory.
//
// Pass order of one frame = RayTracedGGX::OnUpdate + OnRender (RayTracedGGX.cpp:282-353):
//   orc_update_frame -> orc_update_as -> orc_render_visibility -> orc_ray_trace -> orc_denoise -> orc_tone_map
#include <thread>
#include <atomic>
#include <functional>
#include "orc_math.h"
#include "orc_formats.h"
#include "orc_obj.h"
#include "orc_dds.h"
#include "orc_scene.h"
#include "orc_raster.h"
#include "orc_bvh.h"
#include "orc_raytrace.h"
#include "orc_denoise.h"

using namespace orc;

static void parallel_rows(int threads, uint32_t H, const std::function<void(uint32_t)>& fn) {
  if (threads <= 1) { for (uint32_t y = 0; y < H; ++y) fn(y); return; }
  std::atomic<uint32_t> next{0};
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t) pool.emplace_back([&] { for (;;) { const uint32_t y = next.fetch_add(1); if (y >= H) break; fn(y); } });
  for (auto& t : pool) t.join();
}

extern "C" {

void* orc_create(uint32_t width, uint32_t height) {
  Ctx* c = new Ctx();
  c->W = width; c->H = height;
  const size_t n = (size_t)width * height;
  c->vis.assign(n, 0); c->depth.assign(n, 0xFFFFFFu); c->normal.assign(n, 0); c->velocity.assign(n, 0);
  c->refl.assign(n, 0); c->diff.assign(n, 0); c->backbuffer.assign(n, 0); c->roughMetal.assign(n, 0);
  c->tss[0].assign(n, 0); c->tss[1].assign(n, 0); c->fltRfl.assign(n, 0); c->fltDff.assign(n, 0);
  for (int s = 0; s < 256; ++s) {
    const double phi = 2.0 * 3.14159265358979323846 * (double)s / 256.0;
    c->cosTab[s] = (float)std::cos(phi); c->sinTab[s] = (float)std::sin(phi);
  }
  set_default_materials(*c);
  set_ground_mesh(c->mesh[0]);
  c->invWorld[0] = c->invWorld[1] = identity();
  return c;
}
void orc_destroy(void* h) { delete (Ctx*)h; }
void orc_set_threads(void* h, int threads) { ((Ctx*)h)->threads = threads < 1 ? 1 : threads; }

// ---- inputs --------------------------------------------------------------------------------------
// ObjLoader::Import; two-call protocol: counts first, then copy out.
static ObjMesh g_lastObj;
int orc_obj_import(const char* path, uint32_t* numVerts, uint32_t* numIndices, float* aabb6) {
  if (!obj_import(path, g_lastObj)) return -1;
  *numVerts = (uint32_t)(g_lastObj.verts.size() / 6); *numIndices = (uint32_t)g_lastObj.indices.size();
  if (aabb6) { for (int k = 0; k < 3; ++k) { aabb6[k] = g_lastObj.aabbMin[k]; aabb6[3 + k] = g_lastObj.aabbMax[k]; } }
  return 0;
}
void orc_obj_copy(float* verts, uint32_t* indices) {
  std::memcpy(verts, g_lastObj.verts.data(), g_lastObj.verts.size() * 4);
  std::memcpy(indices, g_lastObj.indices.data(), g_lastObj.indices.size() * 4);
}
int orc_set_mesh(void* h, uint32_t slot, const float* verts, uint32_t nv, const uint32_t* idx, uint32_t ni) {
  if (slot > 1) return -1;
  Mesh& m = ((Ctx*)h)->mesh[slot];
  m.verts.assign(verts, verts + 6 * (size_t)nv); m.idx.assign(idx, idx + ni); m.bvh = Bvh{};
  return 0;
}
void orc_set_pos_scale(void* h, const float* ps) { std::memcpy(((Ctx*)h)->posScale, ps, 16); }
void orc_set_sampler(void* h, int vndf) { ((Ctx*)h)->vndf = vndf != 0; }
void orc_set_normal_weight(int variant) { g_normalWeightVariant = variant; }      // orc_denoise.h normal_weight: 0 exact, 1 libm fp32, 2 round 3's squarings (process-wide)
void orc_set_metallic(void* h, uint32_t mesh, float m) { ((Ctx*)h)->fc.mat.RoughMetals[mesh][1] = m; }   // RayTracer.cpp:244-248
void orc_set_material(void* h, uint32_t mesh, const float* baseColor4, float rough, float metal) {
  Ctx* c = (Ctx*)h; std::memcpy(c->fc.mat.BaseColors[mesh], baseColor4, 16); c->fc.mat.RoughMetals[mesh][0] = rough; c->fc.mat.RoughMetals[mesh][1] = metal;
}
// Environment: DDS file, or raw RGBA16F levels (mip-major, 6 faces per mip).
int orc_set_env_dds(void* h, const char* path) {
  char err[256];
  if (!dds_load_cube(path, ((Ctx*)h)->env, err, sizeof err)) { fprintf(stderr, "oracle: %s\n", err); return -1; }
  return 0;
}
int orc_set_env_rgba16f(void* h, uint32_t size, uint32_t mips, const uint16_t* texels) {
  EnvMap& e = ((Ctx*)h)->env; e.size = size; e.mips = mips; e.level.assign((size_t)mips * 6, {});
  size_t off = 0;
  for (uint32_t m = 0; m < mips; ++m) { const uint32_t s = size >> m ? size >> m : 1; for (int f = 0; f < 6; ++f) { auto& l = e.level[(size_t)m * 6 + f]; l.assign(texels + off, texels + off + (size_t)s * s * 4); off += (size_t)s * s * 4; } }
  return 0;
}
uint64_t orc_env_texel_count(void* h) { const EnvMap& e = ((Ctx*)h)->env; uint64_t n = 0; for (auto& l : e.level) n += l.size() / 4; return n; }
void orc_env_info(void* h, uint32_t* size, uint32_t* mips) { *size = ((Ctx*)h)->env.size; *mips = ((Ctx*)h)->env.mips; }
void orc_env_copy(void* h, uint16_t* out) {   // mip-major, 6 faces per mip
  const EnvMap& e = ((Ctx*)h)->env; size_t off = 0;
  for (auto& l : e.level) { std::memcpy(out + off, l.data(), l.size() * 2); off += l.size(); }
}
void orc_bc6h_decode_block(const uint8_t* block16, uint16_t* outRgb48) { bc6h::decodeBlock(block16, (uint16_t(*)[3])outRgb48); }
void orc_bc6h_decode_block_sf16(const uint8_t* block16, uint16_t* outRgb48) { bc6h::decodeBlock(block16, (uint16_t(*)[3])outRgb48, true); }

// ---- acceleration structure ------------------------------------------------------------------------
void orc_build_as(void* h) { Ctx* c = (Ctx*)h; for (int i = 0; i < 2; ++i) build_bvh(c->mesh[i]); }
// Adopt BVH arrays built elsewhere (the product's LBVH, read back by a test) so that the CPU re-traces the same hierarchy.
int orc_set_bvh(void* h, uint32_t slot, const void* nodes, uint32_t numNodes, const void* tris, uint32_t numTris, int32_t root) {
  if (slot > 1) return -1;
  Bvh& b = ((Ctx*)h)->mesh[slot].bvh;
  b.nodes.assign((const BvhNode*)nodes, (const BvhNode*)nodes + numNodes);
  b.tris.assign((const BvhTri*)tris, (const BvhTri*)tris + numTris);
  b.root = root;
  return 0;
}
void orc_bvh_info(void* h, uint32_t slot, uint32_t* numNodes, uint32_t* numTris, int32_t* root) {
  const Bvh& b = ((Ctx*)h)->mesh[slot].bvh; *numNodes = (uint32_t)b.nodes.size(); *numTris = (uint32_t)b.tris.size(); *root = b.root;
}
void orc_bvh_copy(void* h, uint32_t slot, void* nodes, void* tris) {
  const Bvh& b = ((Ctx*)h)->mesh[slot].bvh;
  std::memcpy(nodes, b.nodes.data(), b.nodes.size() * sizeof(BvhNode)); std::memcpy(tris, b.tris.data(), b.tris.size() * sizeof(BvhTri));
}

// ---- per-frame ------------------------------------------------------------------------------------
// Camera of LoadAssets (RayTracedGGX.cpp:19-23, 262-277): returns view*proj (row-major) for the given eye/focus.
void orc_camera_view_proj(uint32_t width, uint32_t height, const float* eye3, const float* focus3, float* viewProj16) {
  const float aspect = (float)width / (float)height;
  const M4 proj = perspective_fov_lh(0.785398163f, aspect, 1.0f, 1000.0f);
  const M4 view = look_at_lh(f3(eye3[0], eye3[1], eye3[2]), f3(focus3[0], focus3[1], focus3[2]), f3(0.0f, 1.0f, 0.0f));
  const M4 vp = mul(view, proj);
  std::memcpy(viewProj16, vp.m, 64);
}
void orc_update_frame(void* h, const float* eye3, const float* viewProj16, float timeStep) {
  M4 vp; std::memcpy(vp.m, viewProj16, 64);
  update_frame(*(Ctx*)h, eye3, vp, timeStep);
}
// Use constants produced elsewhere (e.g. by the product's host code) instead of orc_update_frame.
void orc_set_frame_constants(void* h, const void* fc768) { std::memcpy(&((Ctx*)h)->fc, fc768, sizeof(FrameConstants)); }
void orc_get_frame_constants(void* h, void* fc768) { std::memcpy(fc768, &((Ctx*)h)->fc, sizeof(FrameConstants)); }
void orc_halton(void* h, float* xy) { incremental_halton(*(Ctx*)h, xy[0], xy[1]); }
void orc_update_as(void* h) { update_as(*(Ctx*)h); }
void orc_get_inv_worlds(void* h, float* out32) { Ctx* c = (Ctx*)h; std::memcpy(out32, c->invWorld[0].m, 64); std::memcpy(out32 + 16, c->invWorld[1].m, 64); }
void orc_transform_sh(void* h) { transform_sh(*(Ctx*)h); }
void orc_set_sh(void* h, const float* sh27) { std::memcpy(((Ctx*)h)->sh, sh27, 108); }
void orc_render_visibility(void* h) { render_visibility(*(Ctx*)h); }
// Adopt a visibility/depth pair produced elsewhere (to test later passes in isolation).
void orc_set_visibility(void* h, const uint32_t* vis, const uint32_t* depth) {
  Ctx* c = (Ctx*)h; const size_t n = (size_t)c->W * c->H; c->vis.assign(vis, vis + n); c->depth.assign(depth, depth + n);
}
uint64_t orc_ray_trace(void* h) {
  Ctx* c = (Ctx*)h;
  std::atomic<uint64_t> rays{0};
  parallel_rows(c->threads, c->H, [&](uint32_t y) { uint64_t r = 0; for (uint32_t x = 0; x < c->W; ++x) r += raygen_pixel(*c, x, y); rays += r; });
  c->rayCount = rays.load();
  return c->rayCount;
}
void orc_denoise(void* h) {   // Denoiser::Denoise (Denoiser.cpp:66-75)
  Ctx* c = (Ctx*)h;
  c->frameParity ^= 1u;
  std::vector<uint64_t>& scratch = c->tss[c->frameParity];
  const std::vector<uint64_t>& hist = c->tss[c->frameParity ^ 1u];
  const int T = c->threads; const uint32_t W = c->W, H = c->H;
  parallel_rows(T, H, [&](uint32_t y) { for (uint32_t x = 0; x < W; ++x) spatial_refl_pixel(*c, (int)x, (int)y, false, scratch); });
  parallel_rows(T, H, [&](uint32_t y) { for (uint32_t x = 0; x < W; ++x) spatial_refl_pixel(*c, (int)x, (int)y, true, scratch); });
  parallel_rows(T, H, [&](uint32_t y) { for (uint32_t x = 0; x < W; ++x) spatial_diff_pixel(*c, (int)x, (int)y, false, scratch); });
  parallel_rows(T, H, [&](uint32_t y) { for (uint32_t x = 0; x < W; ++x) spatial_diff_pixel(*c, (int)x, (int)y, true, scratch); });
  // temporal reads FLT_DFF, TSS[!p] and velocity, writes TSS[p] (which was the scratch): needs a separate output
  std::vector<uint64_t> out((size_t)W * H);
  parallel_rows(T, H, [&](uint32_t y) { for (uint32_t x = 0; x < W; ++x) temporal_pixel(*c, (int)x, (int)y, hist, out); });
  scratch.swap(out);
}
void orc_tone_map(void* h) {   // Denoiser::ToneMap (Denoiser.cpp:77-103)
  Ctx* c = (Ctx*)h;
  const std::vector<uint64_t>& src = c->tss[c->frameParity];
  parallel_rows(c->threads, c->H, [&](uint32_t y) { for (uint32_t x = 0; x < c->W; ++x) tonemap_pixel(*c, (int)x, (int)y, src); });
}
// Single passes for isolated parity tests: 0 H_Refl, 1 V_Refl, 2 H_Diff, 3 V_Diff, 4 Temporal (parity must be set by caller)
void orc_flip_parity(void* h) { ((Ctx*)h)->frameParity ^= 1u; }
uint32_t orc_get_parity(void* h) { return ((Ctx*)h)->frameParity; }

// ---- buffers --------------------------------------------------------------------------------------
void* orc_buffer(void* h, int id, uint64_t* bytes) {
  Ctx* c = (Ctx*)h;
  switch (id) {
    case BUF_VISIBILITY: *bytes = c->vis.size() * 4; return c->vis.data();
    case BUF_DEPTH: *bytes = c->depth.size() * 4; return c->depth.data();
    case BUF_NORMAL: *bytes = c->normal.size() * 4; return c->normal.data();
    case BUF_ROUGH_METAL: *bytes = c->roughMetal.size() * 2; return c->roughMetal.data();
    case BUF_VELOCITY: *bytes = c->velocity.size() * 4; return c->velocity.data();
    case BUF_RT_REFL: *bytes = c->refl.size() * 4; return c->refl.data();
    case BUF_RT_DIFF: *bytes = c->diff.size() * 4; return c->diff.data();
    case BUF_TSS0: *bytes = c->tss[0].size() * 8; return c->tss[0].data();
    case BUF_TSS1: *bytes = c->tss[1].size() * 8; return c->tss[1].data();
    case BUF_FLT_RFL: *bytes = c->fltRfl.size() * 8; return c->fltRfl.data();
    case BUF_FLT_DFF: *bytes = c->fltDff.size() * 8; return c->fltDff.data();
    case BUF_BACKBUFFER: *bytes = c->backbuffer.size() * 4; return c->backbuffer.data();
    case BUF_SH_COEFFS: *bytes = 108; return c->sh;
    default: *bytes = 0; return nullptr;
  }
}

// ---- ray queries for tests -------------------------------------------------------------------------
// rays: n x {ox,oy,oz,dx,dy,dz,tmin,tmax}; out: n x {t, inst, prim, b1, b2, valid} as 6 floats (ids bit-cast)
void orc_trace_rays(void* h, const float* rays, uint32_t n, int brute, float* out) {
  Ctx* c = (Ctx*)h;
  parallel_rows(c->threads, n, [&](uint32_t i) {
    const float* r = rays + 8 * (size_t)i;
    const Hit hit = brute ? trace_brute(*c, f3(r[0], r[1], r[2]), f3(r[3], r[4], r[5]), r[6], r[7])
                          : trace_closest(*c, f3(r[0], r[1], r[2]), f3(r[3], r[4], r[5]), r[6], r[7]);
    float* o = out + 6 * (size_t)i;
    o[0] = hit.t; o[1] = u2f(hit.inst); o[2] = u2f(hit.prim); o[3] = hit.b1; o[4] = hit.b2; o[5] = hit.valid ? 1.0f : 0.0f;
  });
}

// traversal statistics of a ray list on the calling thread: out = {node visits, leaf tests, max stack depth}
void orc_trace_stats(void* h, const float* rays, uint32_t n, uint64_t* out3) {
  Ctx* c = (Ctx*)h;
  g_tstats = TraverseStats{};
  for (uint32_t i = 0; i < n; ++i) { const float* r = rays + 8 * (size_t)i; trace_closest(*c, f3(r[0], r[1], r[2]), f3(r[3], r[4], r[5]), r[6], r[7]); }
  out3[0] = g_tstats.nodes; out3[1] = g_tstats.leaves; out3[2] = (uint64_t)g_tstats.maxStack;
}

// statistics accumulated on the calling thread (use orc_set_threads(h, 1) so that passes run on it)
void orc_tstats_reset(void) { g_tstats = TraverseStats{}; }
void orc_tstats_get(uint64_t* out3) { out3[0] = g_tstats.nodes; out3[1] = g_tstats.leaves; out3[2] = (uint64_t)g_tstats.maxStack; }

// ---- small known-answer probes ----------------------------------------------------------------------
uint32_t orc_rng(uint32_t seed) { return rng(seed); }
void orc_sample_param(uint32_t x, uint32_t y, uint32_t W, uint32_t frameIndex, uint32_t* s, float* xi2) {
  const SampleParam p = get_sample_param(x, y, W, frameIndex); *s = p.s; xi2[0] = p.x; xi2[1] = p.y;
}
uint32_t orc_pack_r11g11b10f(const float* rgb) { return pack_r11g11b10f(rgb[0], rgb[1], rgb[2]); }
void orc_unpack_r11g11b10f(uint32_t p, float* rgb) { unpack_r11g11b10f(p, rgb); }
uint16_t orc_f32_to_f16(float f) { return f32_to_f16(f); }
float orc_f16_to_f32(uint16_t h) { return f16_to_f32(h); }
uint32_t orc_pack_r10g10b10a2(const float* v) { return pack_r10g10b10a2(v[0], v[1], v[2], v[3]); }
void orc_environment(void* h, const float* dir3, float level, float* rgb) {
  const float3 c = environment(*(Ctx*)h, f3(dir3[0], dir3[1], dir3[2]), level); rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
}
void orc_matrix_inverse(const float* m16, float* out16) { M4 a; std::memcpy(a.m, m16, 64); const M4 r = inverse(a); std::memcpy(out16, r.m, 64); }

}  // extern "C"
