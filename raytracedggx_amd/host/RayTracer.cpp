#include "RayTracer.h"

#include <cstdio>
#include <cstring>

#include "DDSLoader.h"
#include "ObjLoader.h"

void HaltonSequence::Next(float& x, float& y) {
  // base 2: the bits that change between n and n+1 are flipped from the least significant one
  uint32_t before = m_base2;
  uint32_t changed = before ^ (m_base2 + 1);
  ++m_base2;
  for (float step = 0.5f; changed; step *= 0.5f, changed >>= 1, before >>= 1) m_x += (before & 1u) ? -step : step;
  // base 3: two bits per ternary digit; a digit that reaches 3 wraps to 0 and carries
  ++m_base3;
  uint32_t digitMask = 0x3u, digitOne = 0x1u;
  float step = 1.0f / 3.0f;
  while ((m_base3 & digitMask) == digitMask) {
    m_base3 += digitOne;
    m_y += -2.0f * step;
    digitMask <<= 2; digitOne <<= 2; step *= 1.0f / 3.0f;
  }
  m_y += step;
  x = m_x; y = m_y;
}

RayTracer::RayTracer() { std::memset(m_worldViewProjs, 0, sizeof m_worldViewProjs); }

RayTracer::~RayTracer() { if (m_ctx) rtggx_destroy(m_ctx); }

bool RayTracer::check(int rc, const char* what) {
  if (rc == 0) return true;
  m_error = std::string(what) + ": " + rtggx_last_error();
  std::fprintf(stderr, "RayTracer: %s\n", m_error.c_str());
  return false;
}

bool RayTracer::Init(uint32_t width, uint32_t height, const char* fileName, const char* envFileName,
                     const float posScale[4], int device) {
  m_width = width; m_height = height;
  std::memcpy(m_posScale, posScale, sizeof m_posScale);
  if (!check(rtggx_create(&m_ctx, width, height, device), "rtggx_create")) return false;   // render targets + ground mesh + materials

  // Load inputs (RayTracer.cpp:83-86)
  ObjLoader objLoader;
  if (!objLoader.Import(fileName, true, true)) { m_error = std::string("cannot import ") + fileName; std::fprintf(stderr, "RayTracer: %s\n", m_error.c_str()); return false; }
  m_numVerts = objLoader.GetNumVertices(); m_numIndices = objLoader.GetNumIndices();
  m_modelVerts.assign(reinterpret_cast<const float*>(objLoader.GetVertices()), reinterpret_cast<const float*>(objLoader.GetVertices()) + 6 * (size_t)m_numVerts);
  if (!check(rtggx_set_mesh(m_ctx, MODEL_OBJ, reinterpret_cast<const float*>(objLoader.GetVertices()), m_numVerts,
                            objLoader.GetIndices(), m_numIndices), "rtggx_set_mesh")) return false;

  // Load input image (RayTracer.cpp:143-150)
  DDS::Loader textureLoader;
  DDS::CubeImage cube;
  std::string err;
  if (!textureLoader.LoadCubeFromFile(envFileName, cube, err)) { m_error = err; std::fprintf(stderr, "RayTracer: %s\n", err.c_str()); return false; }
  if (!check(rtggx_set_env(m_ctx, cube.format, cube.size, cube.mips, cube.payload.data(), cube.payload.size()), "rtggx_set_env")) return false;
  return true;
}

bool RayTracer::BuildAccelerationStructures() { return check(rtggx_build_as(m_ctx), "rtggx_build_as"); }

bool RayTracer::Postinit() { return check(rtggx_sync(m_ctx), "rtggx_sync"); }

void RayTracer::SetMetallic(uint32_t meshIdx, float metallic) { check(rtggx_set_metallic(m_ctx, meshIdx, metallic), "rtggx_set_metallic"); }

void RayTracer::SetSampler(bool vndf) { check(rtggx_set_sampler(m_ctx, vndf ? 1 : 0), "rtggx_set_sampler"); }

void RayTracer::SetAsyncCompute(bool asyncCompute) { check(rtggx_set_async_compute(m_ctx, asyncCompute ? 1 : 0), "rtggx_set_async_compute"); }

void RayTracer::UpdateFrame(uint8_t frameIndex, const xm::Float3& eyePt, const xm::Matrix& viewProj, float timeStep) {
  (void)frameIndex;   // the constant-buffer slot ring lives inside librtggx
  using namespace xm;
  float hx, hy;
  m_halton.Next(hx, hy);
  const float projBias[2] = {(hx * 2.0f - 1.0f) / (float)m_width, (hy * 2.0f - 1.0f) / (float)m_height};

  RtggxFrameConstants& cb = m_constants;
  {
    const Matrix projToWorld = Inverse(viewProj);
    StoreFloat4x4(cb.rayGen.ProjToWorld, Transpose(projToWorld));
    cb.rayGen.EyePt[0] = eyePt.x; cb.rayGen.EyePt[1] = eyePt.y; cb.rayGen.EyePt[2] = eyePt.z; cb.rayGen.EyePt[3] = 0.0f;
    cb.rayGen.ProjBias[0] = projBias[0]; cb.rayGen.ProjBias[1] = projBias[1];
    cb.rayGen.pad[0] = cb.rayGen.pad[1] = 0.0f;
  }
  {
    m_angle += 16.0f * timeStep * 3.141592654f / 180.0f;
    const Matrix rot = RotationY(m_angle);
    const Matrix worlds[NUM_MESH] = {
      Scaling(10.0f, 0.5f, 10.0f) * Translation(0.0f, -0.5f, 0.0f),
      Scaling(m_posScale[3], m_posScale[3], m_posScale[3]) * rot * Translation(m_posScale[0], m_posScale[1], m_posScale[2])};
    for (uint32_t i = 0; i < NUM_MESH; ++i) {
      float wvp[16];
      StoreFloat4x4(wvp, Transpose(worlds[i] * viewProj));
      // m_worldViewProjs is never initialised by the reference's constructor; first frame: prev = current
      std::memcpy(cb.global.WorldViewProjsPrev[i], m_hasPrev ? m_worldViewProjs[i] : wvp, 64);
      std::memcpy(cb.global.WorldViewProjs[i], wvp, 64);
      StoreFloat3x4(cb.global.Worlds[i], worlds[i]);
      std::memcpy(m_worldViewProjs[i], wvp, 64);
      std::memcpy(cb.perObject[i].WorldViewProj, wvp, 64);
      cb.perObject[i].ProjBias[0] = projBias[0]; cb.perObject[i].ProjBias[1] = projBias[1];
      cb.perObject[i].pad[0] = cb.perObject[i].pad[1] = 0.0f;
    }
    StoreFloat3x4(cb.global.WorldITs0, Identity());
    StoreFloat3x4(cb.global.WorldIT1, rot, 11);
    m_hasPrev = true;
    cb.global.FrameIndex = m_frameCounter++;
    m_frameCounter %= 256u;
  }
  std::memset(&cb.material, 0, sizeof cb.material);   // CBMaterial is persistent on the device (rtggx_set_material)
  if (m_ctx) check(rtggx_update_frame(m_ctx, &cb), "rtggx_update_frame");
}

void RayTracer::TransformSH() { check(rtggx_transform_sh(m_ctx), "rtggx_transform_sh"); }

void RayTracer::Render(uint8_t frameIndex) {
  RenderVisibility(frameIndex);
  check(rtggx_ray_trace(m_ctx), "rtggx_ray_trace");
}

bool RayTracer::UpdateMesh(const float* vertices, uint32_t numVertices) { return check(rtggx_refit_as(m_ctx, MODEL_OBJ, vertices, numVertices), "rtggx_refit_as"); }

void RayTracer::UpdateAccelerationStructure(uint8_t) { check(rtggx_update_as(m_ctx), "rtggx_update_as"); }

void RayTracer::RenderVisibility(uint8_t, bool) { check(rtggx_render_visibility(m_ctx), "rtggx_render_visibility"); }

void RayTracer::RayTrace(uint8_t) { check(rtggx_ray_trace(m_ctx), "rtggx_ray_trace"); }
