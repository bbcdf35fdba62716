// Host-side mirror of the reference's Denoiser (RayTracedGGX/Content/Denoiser.h:15-21,
// Denoiser.cpp:21-103): Init / Denoise / ToneMap forward to librtggx.  The render targets of
// Denoiser::Init (TemporalSSOut0/1, FilteredOut, FilteredOut1, all RGBA16F) are created together
// with the RayTracer's in rtggx_create; Init only borrows the context, as the reference's Denoiser
// borrows the RayTracer's textures by pointer (Denoiser.cpp:32-34).
#pragma once
#include <cstdint>
#include "../../include/rtggx.h"

class Denoiser {
 public:
  Denoiser() = default;
  virtual ~Denoiser() = default;

  bool Init(rtggx_context* context, uint32_t width, uint32_t height);
  void Denoise(bool useSharedMem = false, bool asyncCompute = false);
  void ToneMap();

 protected:
  rtggx_context* m_ctx = nullptr;   // not owned
  uint32_t m_width = 0, m_height = 0;
};
