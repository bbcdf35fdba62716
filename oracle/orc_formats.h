// ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked, imported or called by the product path.
// Texel format conversions of the render targets the reference creates
// (RayTracer.cpp:91-114, Denoiser.cpp:47-56).  The conversion rules are D3D's, not the
// reference's; where D3D leaves the rounding to the implementation
// the choice made here is: float -> smaller float rounds to nearest even; finite values above
// the largest R11G11B10 value saturate to it; UNORM = floor(clamp(x,0,1) * (2^n-1) + 0.5).
#pragma once
#include <cstdint>
#include <cstring>
#include <cmath>

namespace orc {

static inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

// ---- IEEE binary16 ----------------------------------------------------------------------------
static inline uint16_t f32_to_f16(float f) {
  const uint32_t u = f2u(f);
  const uint32_t sign = (u >> 16) & 0x8000u;
  uint32_t a = u & 0x7FFFFFFFu;
  if (a > 0x7F800000u) return (uint16_t)(sign | 0x7E00u);             // NaN
  if (a >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);            // >= 65520 rounds to inf (inf too)
  if (a < 0x38800000u) {                                              // below 2^-14: half denormal
    if (a < 0x33000000u) return (uint16_t)sign;                       // < 2^-25 rounds to zero
    const uint32_t e = a >> 23;
    const uint32_t mant = (a & 0x7FFFFFu) | 0x800000u;
    const uint32_t shift = 126u - e;                                  // 14..24: mant * 2^(e-150) -> units of 2^-24
    const uint32_t q = mant >> shift, rem = mant & ((1u << shift) - 1u), half = 1u << (shift - 1);
    uint32_t r = q;
    if (rem > half || (rem == half && (q & 1u))) ++r;
    return (uint16_t)(sign | r);
  }
  a -= 0x38000000u;                                                   // rebias 127 -> 15
  a += 0xFFFu + ((a >> 13) & 1u);                                     // round to nearest even
  return (uint16_t)(sign | (a >> 13));
}
static inline float f16_to_f32(uint16_t h) {
  const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  const uint32_t e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
  if (e == 0) {
    if (m == 0) return u2f(sign);
    const float v = (float)m * 5.9604644775390625e-8f;                // m * 2^-24, exact
    return sign ? -v : v;
  }
  if (e == 31) return u2f(sign | 0x7F800000u | (m << 13));
  return u2f(sign | ((e + 112u) << 23) | (m << 13));
}

// ---- unsigned small floats of R11G11B10_FLOAT (5-bit exponent, 6- or 5-bit mantissa) ----------
static inline uint32_t f32_to_ufloat(float f, int mbits) {
  const uint32_t u = f2u(f);
  const uint32_t expMax = 31u << mbits;
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return expMax | 1u;            // NaN
  if (u & 0x80000000u) return 0;                                      // negatives (and -0, -inf) clamp to 0
  if (u == 0x7F800000u) return expMax;                                // +inf
  const uint32_t maxBits = 0x47000000u | (((1u << mbits) - 1u) << (23 - mbits));  // largest finite value
  if (u > maxBits) return expMax - 1u;                                // saturate
  const int shift = 23 - mbits;
  if (u < 0x38800000u) {                                              // denormal in the target format
    const uint32_t e = u >> 23;
    if (e < 127u - 15u - (uint32_t)mbits - 1u) return 0;              // far below half the smallest denormal
    const uint32_t mant = (u & 0x7FFFFFu) | 0x800000u;
    const uint32_t sh = (uint32_t)shift + (113u - e);
    if (sh > 31u) return 0;
    const uint32_t q = mant >> sh, rem = mant & ((1u << sh) - 1u), half = 1u << (sh - 1);
    uint32_t r = q;
    if (rem > half || (rem == half && (q & 1u))) ++r;
    return r;
  }
  uint32_t a = u - 0x38000000u;
  a += ((1u << (shift - 1)) - 1u) + ((a >> shift) & 1u);
  return a >> shift;
}
static inline float ufloat_to_f32(uint32_t v, int mbits) {
  const uint32_t e = v >> mbits, m = v & ((1u << mbits) - 1u);
  if (e == 0) return (float)m * u2f((127u - 14u - (uint32_t)mbits) << 23);   // m * 2^(-14-mbits)
  if (e == 31) return u2f(0x7F800000u | (m << (23 - mbits)));
  return u2f(((e + 112u) << 23) | (m << (23 - mbits)));
}
static inline uint32_t pack_r11g11b10f(float r, float g, float b) {
  return f32_to_ufloat(r, 6) | (f32_to_ufloat(g, 6) << 11) | (f32_to_ufloat(b, 5) << 22);
}
static inline void unpack_r11g11b10f(uint32_t p, float* rgb) {
  rgb[0] = ufloat_to_f32(p & 0x7FFu, 6);
  rgb[1] = ufloat_to_f32((p >> 11) & 0x7FFu, 6);
  rgb[2] = ufloat_to_f32(p >> 22, 5);
}

// ---- UNORM ------------------------------------------------------------------------------------
static inline uint32_t f32_to_unorm(float x, uint32_t maxv) {
  if (!(x > 0.0f)) return 0;                                          // NaN and negatives -> 0
  if (x >= 1.0f) return maxv;
  return (uint32_t)(x * (float)maxv + 0.5f);
}
static inline uint32_t pack_r10g10b10a2(float x, float y, float z, float w) {
  return f32_to_unorm(x, 1023) | (f32_to_unorm(y, 1023) << 10) | (f32_to_unorm(z, 1023) << 20) | (f32_to_unorm(w, 3) << 30);
}
static inline void unpack_r10g10b10a2(uint32_t p, float* v) {
  v[0] = (float)(p & 1023u) / 1023.0f;
  v[1] = (float)((p >> 10) & 1023u) / 1023.0f;
  v[2] = (float)((p >> 20) & 1023u) / 1023.0f;
  v[3] = (float)(p >> 30) / 3.0f;
}
static inline uint16_t pack_r8g8(float x, float y) {
  return (uint16_t)(f32_to_unorm(x, 255) | (f32_to_unorm(y, 255) << 8));
}
// R16G16_FLOAT store (velocity).  The sign of a zero is canonicalised to +0: no consumer can observe
// it, and D3D does not promise -0/+0 distinction through a typed store.
static inline uint32_t pack_r16g16f(float x, float y) {
  uint32_t hx = f32_to_f16(x), hy = f32_to_f16(y);
  if ((hx & 0x7FFFu) == 0) hx = 0;
  if ((hy & 0x7FFFu) == 0) hy = 0;
  return hx | (hy << 16);
}
static inline uint64_t pack_rgba16f(float r, float g, float b, float a) {
  return (uint64_t)f32_to_f16(r) | ((uint64_t)f32_to_f16(g) << 16) | ((uint64_t)f32_to_f16(b) << 32) | ((uint64_t)f32_to_f16(a) << 48);
}
static inline void unpack_rgba16f(uint64_t p, float* v) {
  v[0] = f16_to_f32((uint16_t)p); v[1] = f16_to_f32((uint16_t)(p >> 16));
  v[2] = f16_to_f32((uint16_t)(p >> 32)); v[3] = f16_to_f32((uint16_t)(p >> 48));
}
static inline uint32_t pack_rgba8(float r, float g, float b, float a) {
  return f32_to_unorm(r, 255) | (f32_to_unorm(g, 255) << 8) | (f32_to_unorm(b, 255) << 16) | (f32_to_unorm(a, 255) << 24);
}

}  // namespace orc
