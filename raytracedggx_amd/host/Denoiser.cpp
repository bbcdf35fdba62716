#include "Denoiser.h"

#include <cstdio>

bool Denoiser::Init(rtggx_context* context, uint32_t width, uint32_t height) {
  m_ctx = context; m_width = width; m_height = height;
  return context != nullptr;
}

// Denoiser::Denoise (Denoiser.cpp:66-75): flips the frame parity, then reflection H,V -> diffuse H,V -> temporal.
void Denoiser::Denoise(bool useSharedMem, bool) {
  if (rtggx_denoise(m_ctx, useSharedMem ? 1 : 0) != 0) std::fprintf(stderr, "Denoiser: %s\n", rtggx_last_error());
}

// Denoiser::ToneMap (Denoiser.cpp:77-103)
void Denoiser::ToneMap() {
  if (rtggx_tone_map(m_ctx) != 0) std::fprintf(stderr, "Denoiser: %s\n", rtggx_last_error());
}
