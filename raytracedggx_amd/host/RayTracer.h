// Host-side mirror of the reference's RayTracer (RayTracedGGX/Content/RayTracer.h:24-45): same
// public method names and argument meaning, minus the D3D12 handles.  Every method forwards to one
// entry point of librtggx (include/rtggx.h); the only arithmetic kept on the host is what the
// reference keeps there: the per-frame constants of UpdateFrame (RayTracer.cpp:250-305).
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "../../include/rtggx.h"
#include "XMath.h"

// XUSG::IncrementalHalton (declared RayTracedGGX/XUSG/Advanced/XUSGAdvanced.h:834; body in the
// closed XUSG.dll).  n-th call = (radical_inverse_2(n), radical_inverse_3(n)), accumulated
// incrementally in fp32 -- algorithm and known answers: SURVEY.md row H4 / Appendix F.
class HaltonSequence {
 public:
  void Next(float& x, float& y);
 private:
  uint32_t m_base2 = 0, m_base3 = 0;
  float m_x = 0.0f, m_y = 0.0f;
};

class RayTracer {
 public:
  enum MeshIndex : uint32_t { GROUND, MODEL_OBJ, NUM_MESH };
  static const uint8_t FrameCount = 3;

  RayTracer();
  virtual ~RayTracer();

  // Loads the mesh (OBJ) and the environment (DDS cube), creates the render targets and uploads
  // everything.  posScale = (x, y, z, scale) of the model instance.
  bool Init(uint32_t width, uint32_t height, const char* fileName, const char* envFileName,
            const float posScale[4], int device = 0);
  bool BuildAccelerationStructures();
  bool Postinit();

  void SetMetallic(uint32_t meshIdx, float metallic);
  // m_asyncCompute of the sample (RayTracedGGX.h:118): true = the multi-stream frame (its two queues), false = one stream in
  // submission order (its single command list, RayTracedGGX.cpp:513-556)
  void SetAsyncCompute(bool asyncCompute);
  void SetSampler(bool vndf);            // rtggx_set_sampler: visible-normal (Heitz 2018) sampling of the reflection lobe, opt-in (-vndf)
  void UpdateFrame(uint8_t frameIndex, const xm::Float3& eyePt, const xm::Matrix& viewProj, float timeStep);
  void TransformSH();
  void Render(uint8_t frameIndex);
  void UpdateAccelerationStructure(uint8_t frameIndex);
  // Deforming model: new vertices {Pos, Nrm} for the unchanged topology of the loaded mesh.  The acceleration
  // structure is refitted on the device at the start of the next frame, asynchronously (rtggx_refit_as).
  bool UpdateMesh(const float* vertices, uint32_t numVertices);
  const std::vector<float>& GetModelVertices() const { return m_modelVerts; }       // as imported: 6 floats per vertex
  void RenderVisibility(uint8_t frameIndex, bool asyncCompute = false);
  void RayTrace(uint8_t frameIndex);

  rtggx_context* GetContext() const { return m_ctx; }
  const RtggxFrameConstants& GetFrameConstants() const { return m_constants; }
  uint32_t GetNumModelVertices() const { return m_numVerts; }
  uint32_t GetNumModelIndices() const { return m_numIndices; }
  const std::string& GetLastError() const { return m_error; }

 protected:
  bool check(int rc, const char* what);

  rtggx_context* m_ctx = nullptr;
  uint32_t m_width = 0, m_height = 0;
  float m_posScale[4] = {0.0f, 0.0f, 0.0f, 1.0f};
  uint32_t m_numVerts = 0, m_numIndices = 0;
  std::vector<float> m_modelVerts;

  // state UpdateFrame keeps between frames (statics / members in the reference)
  HaltonSequence m_halton;
  float m_angle = 0.0f;            // RayTracer.cpp:270
  uint32_t m_frameCounter = 0;     // s_frameIndex, RayTracer.cpp:282
  bool m_hasPrev = false;
  float m_worldViewProjs[NUM_MESH][16];   // RayTracer.h:131
  RtggxFrameConstants m_constants{};
  std::string m_error;
};
