// ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked, imported or called by the product path.
// CPU restatement (scalar C++, fp32, no FMA contraction) of the vector/matrix arithmetic the
// reference's hot path relies on.  HLSL intrinsics are restated from their public definitions;
// DirectXMath (not in the reference tree) from its public definitions.
//
// Conventions fixed here (DESIGN.md "numeric conventions"):
//   * every product/sum is individually rounded to fp32, evaluated left to right;
//   * normalize(v) = v * (1 / sqrt(dot(v,v)))   (one IEEE divide, one IEEE sqrt);
//   * mul(v, M) = ((v.x*M[0][j] + v.y*M[1][j]) + v.z*M[2][j]) + v.w*M[3][j]   (row vector, HLSL mul(v,M));
//   * trigonometric constants for matrices come from double-precision libm rounded once to fp32.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };

static inline float3 f3(float x, float y, float z) { return {x, y, z}; }
// (the same text as csrc/rtggx_device.h -- written down twice on purpose: the oracle includes nothing of the product)
// exp2 / log2 of the shading path, part of the numeric contract: the HLSL intrinsics are ~1-ulp hardware functions, libm's are (almost)
// correctly rounded, the device library's are 1-ulp again -- three different last bits.  Both sides of the parity check evaluate THIS
// text instead: range reduction and a fixed polynomial in fp32 (IEEE add / multiply / divide only, no contraction: the same bits on the
// host and on the device), accurate to ~1.5 ulp -- as close to the reference's hardware functions as those are to each other.
// (round 3: with these the raw traced images are compared bit for bit; until then a word was allowed to differ by one code.)
static inline float exp2Contract(float x) {
  if (!(x > -126.0f)) return x != x ? x : 0.0f;      // (results below the normal range: zero -- the shading path asks for 2^(-9.28 NoV))
  if (!(x < 128.0f)) return x != x ? x : __builtin_inff();
  const float n = __builtin_fminf(__builtin_rintf(x), 127.0f);      // (x in (127.5, 128): 2^127 x e^t with t up to 0.693 -- finite, as the true value is; 2^128 does not exist)
  const float t = (x - n) * 0.693147180559945f;      // e^t, |t| <= 0.347 (0.693 in that last half-octave): Taylor to t^8 (next term 2e-10; 1e-7 there)
  float p = 1.0f / 40320.0f;
  p = p * t + 1.0f / 5040.0f; p = p * t + 1.0f / 720.0f; p = p * t + 1.0f / 120.0f; p = p * t + 1.0f / 24.0f;
  p = p * t + 1.0f / 6.0f; p = p * t + 0.5f; p = p * t + 1.0f; p = p * t + 1.0f;
  union { uint32_t u; float f; } s; s.u = (uint32_t)(127 + (int)n) << 23;      // 2^n, n in [-126, 127]
  return p * s.f;
}
static inline float log2Contract(float x) {
  if (!(x > 0.0f)) return x == 0.0f ? -__builtin_inff() : __builtin_nanf("");
  if (!(x < __builtin_inff())) return x;
  union { float f; uint32_t u; } v; v.f = x;
  int e = 0;
  if (v.u < 0x00800000u) { v.f = x * 8388608.0f; e = -23; }      // subnormal: scaled into the normal range
  e += (int)(v.u >> 23) - 127;
  v.u = (v.u & 0x007FFFFFu) | 0x3F800000u;                           // mantissa in [1, 2)
  float m = v.f;
  if (m > 1.41421356f) { m *= 0.5f; e += 1; }                        // [sqrt(1/2), sqrt(2)): |s| <= 0.1716
  const float s = (m - 1.0f) / (m + 1.0f), s2 = s * s;
  float q = 1.0f / 9.0f;                                             // atanh series to s^9 (next term 3e-10)
  q = q * s2 + 1.0f / 7.0f; q = q * s2 + 1.0f / 5.0f; q = q * s2 + 1.0f / 3.0f; q = q * s2 + 1.0f;
  return (float)e + ((2.0f * s) * q) * 1.44269504088896f;
}

static inline float3 operator+(float3 a, float3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline float3 operator-(float3 a, float3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline float3 operator*(float3 a, float3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
static inline float3 operator*(float3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static inline float3 operator*(float s, float3 a) { return {s * a.x, s * a.y, s * a.z}; }
static inline float3 operator/(float3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
static inline float3 operator-(float3 a) { return {-a.x, -a.y, -a.z}; }
static inline float dot(float3 a, float3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline float dot(float2 a, float2 b) { return a.x * b.x + a.y * b.y; }
static inline float3 cross(float3 a, float3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
static inline float3 normalize(float3 v) {
  const float inv = 1.0f / std::sqrt(dot(v, v));
  return v * inv;
}
static inline float length(float3 v) { return std::sqrt(dot(v, v)); }
static inline float saturate(float x) { return std::fmin(std::fmax(x, 0.0f), 1.0f); }
static inline float clampf(float x, float lo, float hi) { return std::fmin(std::fmax(x, lo), hi); }
static inline float lerp(float a, float b, float t) { return a + t * (b - a); }
static inline float3 lerp(float3 a, float3 b, float t) { return a + t * (b - a); }
// HLSL reflect(i, n) = i - 2 * dot(i, n) * n
static inline float3 reflect(float3 i, float3 n) {
  const float k = 2.0f * dot(i, n);
  return i - k * n;
}
// HLSL smoothstep(a, b, x)
static inline float smoothstep(float a, float b, float x) {
  const float t = saturate((x - a) / (b - a));
  return t * t * (3.0f - 2.0f * t);
}
// HLSL float -> uint conversion (ftou): NaN and negatives give 0, large values saturate.
static inline uint32_t ftou(float x) {
  if (!(x > 0.0f)) return 0u;
  if (x >= 4294967296.0f) return 0xFFFFFFFFu;
  return (uint32_t)x;
}

// ---------------------------------------------------------------------------------------------
// 4x4 matrix, DirectXMath conventions: row-major storage, row vectors (v' = v * M), left-handed.
// ---------------------------------------------------------------------------------------------
struct M4 { float m[4][4]; };

static inline M4 identity() {
  M4 r{};
  for (int i = 0; i < 4; ++i) r.m[i][i] = 1.0f;
  return r;
}
// XMMatrixMultiply(A, B) = A * B
static inline M4 mul(const M4& a, const M4& b) {
  M4 r;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      r.m[i][j] = ((a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j]) + a.m[i][2] * b.m[2][j]) + a.m[i][3] * b.m[3][j];
  return r;
}
static inline M4 transpose(const M4& a) {
  M4 r;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = a.m[j][i];
  return r;
}
static inline M4 scaling(float x, float y, float z) {
  M4 r = identity(); r.m[0][0] = x; r.m[1][1] = y; r.m[2][2] = z; return r;
}
static inline M4 translation(float x, float y, float z) {
  M4 r = identity(); r.m[3][0] = x; r.m[3][1] = y; r.m[3][2] = z; return r;
}
// XMMatrixRotationY: [[c,0,-s,0],[0,1,0,0],[s,0,c,0],[0,0,0,1]]
static inline M4 rotation_y(float angle) {
  const float s = (float)std::sin((double)angle), c = (float)std::cos((double)angle);
  M4 r = identity();
  r.m[0][0] = c; r.m[0][2] = -s; r.m[2][0] = s; r.m[2][2] = c;
  return r;
}
// XMMatrixLookAtLH(eye, focus, up)
static inline M4 look_at_lh(float3 eye, float3 focus, float3 up) {
  const float3 z = normalize(focus - eye);
  const float3 x = normalize(cross(up, z));
  const float3 y = cross(z, x);
  M4 r = identity();
  r.m[0][0] = x.x; r.m[0][1] = y.x; r.m[0][2] = z.x;
  r.m[1][0] = x.y; r.m[1][1] = y.y; r.m[1][2] = z.y;
  r.m[2][0] = x.z; r.m[2][1] = y.z; r.m[2][2] = z.z;
  r.m[3][0] = -dot(x, eye); r.m[3][1] = -dot(y, eye); r.m[3][2] = -dot(z, eye);
  return r;
}
// XMMatrixPerspectiveFovLH(fovY, aspect, zn, zf)
static inline M4 perspective_fov_lh(float fovY, float aspect, float zn, float zf) {
  const double half = 0.5 * (double)fovY;
  const float sinFov = (float)std::sin(half), cosFov = (float)std::cos(half);
  const float h = cosFov / sinFov;
  const float w = h / aspect;
  const float range = zf / (zf - zn);
  M4 r{};
  r.m[0][0] = w; r.m[1][1] = h; r.m[2][2] = range; r.m[2][3] = 1.0f; r.m[3][2] = -range * zn;
  return r;
}
// XMMatrixInverse: general 4x4 inverse by cofactors, evaluated in double and rounded once.
static inline M4 inverse(const M4& a) {
  double m[16], inv[16];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) m[i * 4 + j] = (double)a.m[i][j];
  inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
  inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
  inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
  inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
  inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
  inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
  inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
  inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
  inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
  inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
  inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
  inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
  inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
  inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
  inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
  inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
  const double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
  const double rdet = 1.0 / det;
  M4 r;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = (float)(inv[i * 4 + j] * rdet);
  return r;
}
// HLSL mul(float4(p,1), M): row vector times matrix.
static inline float4 mul_point(float3 p, const M4& M) {
  float4 r;
  r.x = ((p.x * M.m[0][0] + p.y * M.m[1][0]) + p.z * M.m[2][0]) + M.m[3][0];
  r.y = ((p.x * M.m[0][1] + p.y * M.m[1][1]) + p.z * M.m[2][1]) + M.m[3][1];
  r.z = ((p.x * M.m[0][2] + p.y * M.m[1][2]) + p.z * M.m[2][2]) + M.m[3][2];
  r.w = ((p.x * M.m[0][3] + p.y * M.m[1][3]) + p.z * M.m[2][3]) + M.m[3][3];
  return r;
}
static inline float4 mul_vec4(float4 p, const M4& M) {
  float4 r;
  r.x = ((p.x * M.m[0][0] + p.y * M.m[1][0]) + p.z * M.m[2][0]) + p.w * M.m[3][0];
  r.y = ((p.x * M.m[0][1] + p.y * M.m[1][1]) + p.z * M.m[2][1]) + p.w * M.m[3][1];
  r.z = ((p.x * M.m[0][2] + p.y * M.m[1][2]) + p.z * M.m[2][2]) + p.w * M.m[3][2];
  r.w = ((p.x * M.m[0][3] + p.y * M.m[1][3]) + p.z * M.m[2][3]) + p.w * M.m[3][3];
  return r;
}
// mul(float3 n, float3x3 M) with the upper-left 3x3 of M.
static inline float3 mul_dir(float3 n, const M4& M) {
  return {(n.x * M.m[0][0] + n.y * M.m[1][0]) + n.z * M.m[2][0],
          (n.x * M.m[0][1] + n.y * M.m[1][1]) + n.z * M.m[2][1],
          (n.x * M.m[0][2] + n.y * M.m[1][2]) + n.z * M.m[2][2]};
}

}  // namespace orc
