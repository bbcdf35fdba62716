// ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked, imported or called by the product path.
// Restatement of the denoiser chain, direct-access path (the parity target):
//   CSSpatial_H_Refl.hlsl:15-50, CSSpatial_V_Refl.hlsl:16-59, CSSpatial_H_Diff.hlsl:15-48,
//   CSSpatial_V_Diff.hlsl:17-59, SpatialFilter.hlsli:57-83, FilterCommon.hlsli:14-71,
//   CSTemporalSS.hlsl (built with _DENOISE_;_ALPHA_AS_ID_, vcxproj:217-218; HALF = fp32),
//   PSToneMap.hlsl:13-41, pass order / ping-pong of Denoiser.cpp:66-75,361-478.
// D3D rules restated: out-of-bounds texel loads return 0; SampleLevel(LINEAR_CLAMP) is a
// bilinear fetch with clamped addresses (full fp32 weights here); typed stores convert with
// orc_formats.h.  The H passes store a float3 into RGBA16F: alpha is written as 0 (never read).
#pragma once
#include <cstdlib>
#include "orc_scene.h"
#include "orc_formats.h"

namespace orc {

struct GTexel { float n[3]; float nw; float rough, metal; float depth; };

static inline GTexel load_g(const Ctx& c, int x, int y) {
  GTexel g{};
  if (x < 0 || y < 0 || x >= (int)c.W || y >= (int)c.H) {   // out of bounds: every channel 0
    g.n[0] = g.n[1] = g.n[2] = -1.0f;                       // 0 * 2 - 1
    return g;
  }
  const size_t i = (size_t)y * c.W + (size_t)x;
  float n[4]; unpack_r10g10b10a2(c.normal[i], n);
  for (int k = 0; k < 3; ++k) g.n[k] = n[k] * 2.0f - 1.0f;
  g.nw = n[3];
  g.rough = (float)(c.roughMetal[i] & 0xFF) / 255.0f; g.metal = (float)(c.roughMetal[i] >> 8) / 255.0f;
  g.depth = (float)c.depth[i] / 16777215.0f;                // D24_UNORM
  return g;
}
static inline void load_rgb11(const std::vector<uint32_t>& b, const Ctx& c, int x, int y, float* rgb) {
  if (x < 0 || y < 0 || x >= (int)c.W || y >= (int)c.H) { rgb[0] = rgb[1] = rgb[2] = 0.0f; return; }
  unpack_r11g11b10f(b[(size_t)y * c.W + (size_t)x], rgb);
}
static inline void load_rgba16(const std::vector<uint64_t>& b, const Ctx& c, int x, int y, float* v) {
  if (x < 0 || y < 0 || x >= (int)c.W || y >= (int)c.H) { v[0] = v[1] = v[2] = v[3] = 0.0f; return; }
  unpack_rgba16f(b[(size_t)y * c.W + (size_t)x], v);
}

// FilterCommon.hlsli
static inline void TM3(float* rgb) { const float l = 1.0f + ((rgb[0] * 0.25f + rgb[1] * 0.5f) + rgb[2] * 0.25f); rgb[0] /= l; rgb[1] /= l; rgb[2] /= l; }     // :14-19
static inline void ITM3(float* rgb) { const float l = 1.0f - ((rgb[0] * 0.25f + rgb[1] * 0.5f) + rgb[2] * 0.25f); rgb[0] /= l; rgb[1] /= l; rgb[2] /= l; }    // :24-27
// pow(max(dot(nC, n), 0), sigma), sigma = 512 or 32 (SpatialFilter.hlsli:62,73; FilterCommon.hlsli:34-37).  HLSL's pow is exp2(y * log2(x))
// on the hardware's 1-ulp functions -- for x next to 1 uncertain in the last few bits -- so no evaluation is "the" reference's; what the
// oracle owes the parity check is one that owes nothing to the product's.  Round 3 had both sides square x nine (five) times -- the same
// 423 ulps of error on both sides, and a temporal pass that turns a 1e-5 difference of its input into 1e-3 of its output made the check
// pass BECAUSE the rounding was shared.  Round 4: the product computes v_exp_f32(sigma * v_log_f32(x)) (16 ulps where the weight counts),
// and the oracle has three evaluations to hold against it (orc_set_normal_weight; the tests run the first two):
//   0  EXACT    the dot product in double (exact for 10-bit normals), the power in double, rounded once -- the true value to half an ulp
//   1  LIBM     what a plain C reading of the HLSL gives: fp32 dot product (two multiply-adds unfused: mul, mul, add, mul, add), std::pow in fp32
//   2  SQUARED  round 3's: fused dot product, nine / five fp32 squarings (kept for measurement: tools/probes/parity_probe.py)
static int g_normalWeightVariant = std::getenv("RTGGX_ORACLE_LIBM_POW") ? 1 : 0;
static inline float normal_weight(const float* a, const float* b, float sigma) {   // :34-37
  if (g_normalWeightVariant == 0) {
    const double d = ((double)a[0] * (double)b[0] + (double)a[1] * (double)b[1]) + (double)a[2] * (double)b[2];
    return (float)std::pow(d > 0.0 ? d : 0.0, (double)sigma);
  }
  if (g_normalWeightVariant == 1) {
    const volatile float m0 = a[0] * b[0], m1 = a[1] * b[1], m2 = a[2] * b[2];      // (volatile: no contraction, whatever the flags)
    const float p = std::fmax((m0 + m1) + m2, 0.0f);
    return std::pow(p, sigma);
  }
  float p = std::fmax(std::fmaf(a[2], b[2], std::fmaf(a[1], b[1], a[0] * b[0])), 0.0f);
  if (sigma != 512.0f && sigma != 32.0f) return std::pow(p, sigma);
  p *= p; p *= p; p *= p; p *= p; p *= p;        // ^32
  if (sigma == 512.0f) { p *= p; p *= p; p *= p; p *= p; }
  return p;
}
static inline float depth_weight(float dc, float d, float sigma) { return std::exp(-std::fabs(dc - d) * dc * sigma); }   // :39-42
static inline float roughness_weight(float rc, float r, float smin, float smax) { return 1.0f - smoothstep(smin, smax, std::fabs(r - rc)); }   // :44-47
static inline int gaussian_radius_from_roughness(float rough, float vx, float vy) {   // :49-52
  return (int)clampf(0.1f * rough * vx, 0.0f, vy * 0.05f);
}
static inline float gaussian(float r, int radius) {   // :59-71
  const float sigma = (float)(radius + 1) / 3.0f;
  const float a = r / sigma;
  return std::exp(-0.5f * a * a);
}
// SpatialFilter.hlsli:57-67
static inline float reflection_weight(const float* nC, const GTexel& g, float rghC, float depthC, float radius, int br) {
  float w = g.nw > 0.0f ? 1.0f : 0.0f;
  w *= gaussian(radius, br);
  w *= normal_weight(nC, g.n, 512.0f);
  w *= depth_weight(depthC, g.depth, 4.0f);
  w *= roughness_weight(rghC, g.rough, 0.0f, 0.5f);
  return w;
}
// SpatialFilter.hlsli:69-75
static inline float diffuse_weight(const float* nC, const GTexel& g, float depthC) {
  float w = normal_weight(nC, g.n, 32.0f);
  w *= depth_weight(depthC, g.depth, 4.0f);
  return w;
}

static const int kRadius = 16;   // SpatialFilter.hlsli:8

// CSSpatial_H_Refl / CSSpatial_V_Refl.  vertical=false: src = raw reflection -> scratch;
// vertical=true: src = scratch -> FilteredOut.
static inline void spatial_refl_pixel(Ctx& c, int x, int y, bool vertical, std::vector<uint64_t>& scratch) {
  const size_t pix = (size_t)y * c.W + (size_t)x;
  const GTexel gc = load_g(c, x, y);
  if (gc.nw <= 0.0f) {
    if (vertical) { float s[3]; load_rgb11(c.refl, c, x, y, s); c.fltRfl[pix] = pack_rgba16f(s[0], s[1], s[2], 0.0f); }   // V :22-26
    return;                                                                                                              // H :19
  }
  const int br = gaussian_radius_from_roughness(gc.rough, (float)c.W, (float)c.H);
  float mu[3] = {0, 0, 0}, wsum = 0.0f;
  for (int i = -kRadius; i <= kRadius; ++i) {
    const int tx = vertical ? x : x + i, ty = vertical ? y + i : y;
    const GTexel g = load_g(c, tx, ty);
    float src[4];
    if (vertical) load_rgba16(scratch, c, tx, ty, src);
    else { load_rgb11(c.refl, c, tx, ty, src); TM3(src); }
    const float w = reflection_weight(gc.n, g, gc.rough, gc.depth, vertical ? (float)i : (float)std::abs(i), br);
    for (int k = 0; k < 3; ++k) mu[k] += src[k] * w;
    wsum += w;
  }
  for (int k = 0; k < 3; ++k) mu[k] /= wsum;
  if (vertical) { ITM3(mu); c.fltRfl[pix] = pack_rgba16f(mu[0], mu[1], mu[2], 1.0f); }
  else scratch[pix] = pack_rgba16f(mu[0], mu[1], mu[2], 0.0f);
}

// CSSpatial_H_Diff / CSSpatial_V_Diff
static inline void spatial_diff_pixel(Ctx& c, int x, int y, bool vertical, std::vector<uint64_t>& scratch) {
  const size_t pix = (size_t)y * c.W + (size_t)x;
  const GTexel gc = load_g(c, x, y);
  if (gc.nw <= 0.0f || gc.metal >= 1.0f) {
    if (vertical) c.fltDff[pix] = c.fltRfl[pix];   // V :24-28 (dest passes through; same format, bit copy)
    return;                                        // H :19
  }
  float mu[3] = {0, 0, 0}, wsum = 0.0f;
  for (int i = -kRadius; i <= kRadius; ++i) {
    const int tx = vertical ? x : x + i, ty = vertical ? y + i : y;
    const GTexel g = load_g(c, tx, ty);
    if (g.nw <= 0.0f || g.metal >= 1.0f) continue;
    float src[4];
    if (vertical) load_rgba16(scratch, c, tx, ty, src);
    else { load_rgb11(c.diff, c, tx, ty, src); TM3(src); }
    const float w = diffuse_weight(gc.n, g, gc.depth);
    for (int k = 0; k < 3; ++k) mu[k] += src[k] * w;
    wsum += w;
  }
  for (int k = 0; k < 3; ++k) mu[k] /= wsum;
  if (vertical) {
    float dest[4]; unpack_rgba16f(c.fltRfl[pix], dest);
    ITM3(mu);
    c.fltDff[pix] = pack_rgba16f(dest[0] + mu[0], dest[1] + mu[1], dest[2] + mu[2], dest[3]);
  } else scratch[pix] = pack_rgba16f(mu[0], mu[1], mu[2], 0.0f);
}

// ---- CSTemporalSS.hlsl --------------------------------------------------------------------------
static inline void rgb_to_ycocg(const float* rgb, float* o) {   // :78-85
  o[0] = (rgb[0] * 1.0f + rgb[1] * 2.0f) + rgb[2] * 1.0f;
  o[1] = (rgb[0] * 2.0f + rgb[1] * 0.0f) + rgb[2] * -2.0f;
  o[2] = (rgb[0] * -1.0f + rgb[1] * 2.0f) + rgb[2] * -1.0f;
}
static inline void ycocg_to_rgb(const float* ycc, float* o) {   // :90-101
  const float y = ycc[0] * 0.25f, co = ycc[1] * 0.25f, cg = ycc[2] * 0.25f;
  o[0] = y + co - cg; o[1] = y + cg; o[2] = y - co - cg;
}
static inline void tss_TM(const float* hdr, float* o) {   // :106-114
  float c[3]; rgb_to_ycocg(hdr, c);
  const float d = 4.0f + c[0];
  o[0] = c[0] / d; o[1] = c[1] / d; o[2] = c[2] / d;
}
static inline void tss_ITM(const float* col, float* o) {   // :119-128
  const float k = 4.0f / (1.0f - col[0]);
  const float c[3] = {col[0] * k, col[1] * k, col[2] * k};
  ycocg_to_rgb(c, o);
}
static inline void load_vel(const Ctx& c, int x, int y, float* v) {
  if (x < 0 || y < 0 || x >= (int)c.W || y >= (int)c.H) { v[0] = v[1] = 0.0f; return; }
  const uint32_t p = c.velocity[(size_t)y * c.W + (size_t)x];
  v[0] = f16_to_f32((uint16_t)p); v[1] = f16_to_f32((uint16_t)(p >> 16));
}
static const int kTexOffsets[8][2] = {{-1, 0}, {1, 0}, {0, -1}, {0, 1}, {-1, -1}, {1, -1}, {1, 1}, {-1, 1}};   // :48-52

static inline void temporal_pixel(Ctx& c, int x, int y, const std::vector<uint64_t>& hist, std::vector<uint64_t>& out) {
  const float W = (float)c.W, H = (float)c.H;
  const float uv[2] = {((float)x + 0.5f) / W, ((float)y + 0.5f) / H};
  float current[4]; load_rgba16(c.fltDff, c, x, y, current);
  // VelocityMax :133-161
  float vmax[2]; load_vel(c, x, y, vmax);
  float speedSq = vmax[0] * vmax[0] + vmax[1] * vmax[1];
  for (int i = 0; i < 4; ++i) {
    float nb[2]; load_vel(c, x + kTexOffsets[i + 4][0], y + kTexOffsets[i + 4][1], nb);
    const float sq = nb[0] * nb[0] + nb[1] * nb[1];
    if (sq > speedSq) { vmax[0] = nb[0]; vmax[1] = nb[1]; speedSq = sq; }
  }
  // history = g_txHistory.SampleLevel(g_smpLinear, uv - velocity, 0)  :259-260
  float history[4];
  {
    const float sx = (uv[0] - vmax[0]) * W - 0.5f, sy = (uv[1] - vmax[1]) * H - 0.5f;
    const float x0 = std::floor(sx), y0 = std::floor(sy);
    const float fx = sx - x0, fy = sy - y0;
    auto cl = [](float v, int hi) { return v < 0.0f ? 0 : (v > (float)hi ? hi : (int)v); };
    const int ix0 = cl(x0, (int)c.W - 1), ix1 = cl(x0 + 1.0f, (int)c.W - 1), iy0 = cl(y0, (int)c.H - 1), iy1 = cl(y0 + 1.0f, (int)c.H - 1);
    float t00[4], t10[4], t01[4], t11[4];
    unpack_rgba16f(hist[(size_t)iy0 * c.W + ix0], t00); unpack_rgba16f(hist[(size_t)iy0 * c.W + ix1], t10);
    unpack_rgba16f(hist[(size_t)iy1 * c.W + ix0], t01); unpack_rgba16f(hist[(size_t)iy1 * c.W + ix1], t11);
    const float w00 = (1.0f - fx) * (1.0f - fy), w10 = fx * (1.0f - fy), w01 = (1.0f - fx) * fy, w11 = fx * fy;
    for (int k = 0; k < 4; ++k) history[k] = ((t00[k] * w00 + t10[k] * w10) + t01[k] * w01) + t11[k] * w11;
  }
  // :262-275
  const float hb[2] = {std::fabs(vmax[0]) * (4.0f * W), std::fabs(vmax[1]) * (4.0f * H)};
  float curHistoryBlur = hb[0] + hb[1];
  float historyBlur = 1.0f - history[3];
  historyBlur = std::fmax(historyBlur, curHistoryBlur);
  history[3] = history[3] * 15.0f + 1.0f;
  float currentTM[4]; tss_TM(current, currentTM); currentTM[3] = current[3];
  float gamma = current[3] <= 0.0f ? 1.0f : clampf(8.0f / historyBlur, 1.0f, 32.0f);   // :280-281 (_DENOISE_)
  // NeighborMinMax :166-236 (_VARIANCE_AABB_, _ALPHA_AS_ID_, _DENOISE_)
  float filtered[4] = {currentTM[0], currentTM[1], currentTM[2], currentTM[3]};
  float nmin[4], nmax[4];
  {
    static const float weights[8] = {0.5f, 0.5f, 0.5f, 0.5f, 0.25f, 0.25f, 0.25f, 0.25f};
    float mu[3] = {currentTM[0], currentTM[1], currentTM[2]};
    const float alpha = currentTM[3];
    float m2[3] = {mu[0] * mu[0], mu[1] * mu[1], mu[2] * mu[2]};
    for (int i = 0; i < 8; ++i) {
      float nraw[4]; load_rgba16(c.fltDff, c, x + kTexOffsets[i][0], y + kTexOffsets[i][1], nraw);
      float nb[4]; tss_TM(nraw, nb); nb[3] = nraw[3];
      for (int k = 0; k < 4; ++k) filtered[k] += nb[k] * weights[i];
      for (int k = 0; k < 3; ++k) { mu[k] += nb[k]; m2[k] += nb[k] * nb[k]; }
    }
    for (int k = 0; k < 4; ++k) filtered[k] /= 4.0f;
    gamma = std::fabs(alpha - filtered[3]) < 1.0f / 255.0f ? gamma : 1.0f;
    float sigma[3];
    for (int k = 0; k < 3; ++k) {
      mu[k] /= 9.0f;
      sigma[k] = std::sqrt(std::fabs(m2[k] / 9.0f - mu[k] * mu[k]));
      const float gs = gamma * sigma[k];
      nmin[k] = std::fmin(mu[k] - gs, filtered[k]);
      nmax[k] = std::fmax(mu[k] + gs, filtered[k]);
    }
    nmin[3] = mu[0] - sigma[0]; nmax[3] = mu[0] + sigma[0];
  }
  curHistoryBlur = saturate(curHistoryBlur);   // :290-291
  historyBlur = saturate(historyBlur);
  float historyTM[3]; tss_TM(history, historyTM);   // :294-299
  for (int k = 0; k < 3; ++k) historyTM[k] = std::fmin(std::fmax(historyTM[k], nmin[k]), nmax[k]);
  const float contrast = nmax[3] - nmin[3];
  const float lumContrastFactor = 32.0f * 4.0f;   // :303-308
  float addAlias = historyBlur * 0.5f + 0.25f;
  addAlias = saturate(addAlias + 1.0f / (1.0f + contrast * lumContrastFactor));
  for (int k = 0; k < 3; ++k) filtered[k] = lerp(filtered[k], currentTM[k], addAlias);   // :311
  const float lumHist = historyTM[0];   // :314-325
  const float distToClamp = std::fmin(std::fabs(nmin[3] - lumHist), std::fabs(nmax[3] - lumHist));
  const float historyAmt = std::fmin(1.0f / history[3] + historyBlur / 8.0f, 1.0f);
  float blend = 0.25f / lerp(8.0f, distToClamp + contrast, historyAmt);
  blend = std::fmin(blend, 0.25f);
  blend = filtered[3] > 0.0f ? blend : 1.0f;
  float mix[3]; for (int k = 0; k < 3; ++k) mix[k] = lerp(historyTM[k], filtered[k], blend);   // :327-329
  float result[3]; tss_ITM(mix, result);
  if (std::isnan(result[0]) || std::isnan(result[1]) || std::isnan(result[2])) tss_ITM(filtered, result);
  const float hw = std::fmin(history[3] / 15.0f, 1.0f - curHistoryBlur);
  out[(size_t)y * c.W + (size_t)x] = pack_rgba16f(result[0], result[1], result[2], hw);   // :335
}

// PSToneMap.hlsl:13-41
static inline void tonemap_pixel(Ctx& c, int x, int y, const std::vector<uint64_t>& src) {
  float col[5][4];
  load_rgba16(src, c, x, y, col[0]); load_rgba16(src, c, x - 1, y, col[1]); load_rgba16(src, c, x + 1, y, col[2]);
  load_rgba16(src, c, x, y - 1, col[3]); load_rgba16(src, c, x, y + 1, col[4]);
  for (int i = 0; i < 5; ++i) for (int k = 0; k < 3; ++k) col[i][k] /= col[i][k] + 0.5f;
  float out[3];
  for (int k = 0; k < 3; ++k) {
    float lap = -4.0f * col[0][k];
    for (int i = 1; i < 5; ++i) lap += col[i][k];
    out[k] = col[0][k] - 0.2f * lap;
  }
  c.backbuffer[(size_t)y * c.W + (size_t)x] = pack_rgba8(out[0], out[1], out[2], col[0][3]);
}

}  // namespace orc
