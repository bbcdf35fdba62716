// Device-side arithmetic shared by the gfx950 kernels of librtggx.
//
// Numeric contract (DESIGN.md "numeric conventions"): the library is compiled with
// -ffp-contract=off, so every fp32 product and sum is rounded on its own, in the order written
// here; divisions and square roots are the correctly rounded HIP defaults.  The formulas follow
// the HLSL of the reference (file:line at each function); HLSL intrinsics are spelled out:
//   normalize(v) = v * (1 / sqrt(dot(v,v))),  reflect(i,n) = i - 2*dot(i,n)*n,
//   lerp(a,b,t) = a + t*(b-a),  mul(v,M) accumulates left to right.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/rtggx.h"

#define RT_DEV __device__ __forceinline__
#define RT_HD __host__ __device__ __forceinline__

namespace rt {

struct f3 { float x, y, z; };
struct f2 { float x, y; };
struct f4 { float x, y, z, w; };

RT_HD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_HD f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_HD f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_HD f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_HD f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
RT_HD f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
RT_HD f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
RT_HD float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
RT_HD f3 cross3(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
RT_DEV f3 normalize3(f3 v) { const float inv = 1.0f / sqrtf(dot3(v, v)); return v * inv; }
RT_DEV float saturatef(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
RT_DEV float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
RT_HD float lerpf(float a, float b, float t) { return a + t * (b - a); }
RT_HD f3 lerp3(f3 a, f3 b, float t) { return a + t * (b - a); }
RT_HD f3 reflect3(f3 i, f3 n) { const float k = 2.0f * dot3(i, n); return i - k * n; }
RT_DEV float smoothstepf(float a, float b, float x) { const float t = saturatef((x - a) / (b - a)); return t * t * (3.0f - 2.0f * t); }

// ---- correctly rounded division / square root in fewer instructions -------------------------------
// `a / b` and sqrtf() expand to 10-16 instructions because they also cover operands near the ends of the exponent range
// (v_div_scale / v_div_fixup, the 2^32 pre-scaling of sqrt).  Where the operands cannot be there (|x| in [1e-30, 1e30] or
// zero) the core of the same sequences returns the same round-to-nearest result; tools/microbench/exactmath.hip checks
// that exhaustively (x/9, sqrt) and on 4e9 random triples (div3Shared) against the compiler's expansion on the GPU.
RT_DEV float divBy9(float x) {              // q = RN(x * RN(1/9)), one residual correction (Markstein); -0 comes back as +0
  const float r9 = 0.111111112f;            // RN(1/9) = 0x3DE38E39
  const float q = x * r9;
  return __builtin_fmaf(__builtin_fmaf(-9.0f, q, x), r9, q);
}
RT_DEV void div3Shared(float n0, float n1, float n2, float d, float& q0, float& q1, float& q2) {   // n0/d, n1/d, n2/d
  const float y0 = __builtin_amdgcn_rcpf(d);
  const float y = __builtin_fmaf(__builtin_fmaf(-d, y0, 1.0f), y0, y0);       // 1/d refined once, shared
  auto quotient = [&](float n) {
    const float a = n * y;
    const float b = __builtin_fmaf(__builtin_fmaf(-d, a, n), y, a);
    return __builtin_fmaf(__builtin_fmaf(-d, b, n), y, b);
  };
  q0 = quotient(n0); q1 = quotient(n1); q2 = quotient(n2);
}
RT_DEV float sqrtRN(float x) {              // v_sqrt_f32 (1 ulp), then pick among s-1ulp, s, s+1ulp by the sign of the residuals
  const float s = __builtin_amdgcn_sqrtf(x);
  const float dn = __uint_as_float(__float_as_uint(s) - 1u), up = __uint_as_float(__float_as_uint(s) + 1u);
  const float rdn = __builtin_fmaf(-dn, s, x), rup = __builtin_fmaf(-up, s, x);
  float r = rdn <= 0.0f ? dn : s;
  r = rup > 0.0f ? up : r;
  return r;
}

#define RT_MATH RT_HD
// exp2 / log2 of the shading path, part of the numeric contract: the HLSL intrinsics are ~1-ulp hardware functions, libm's are (almost)
// correctly rounded, the device library's are 1-ulp again -- three different last bits.  Both sides of the parity check evaluate THIS
// text instead: range reduction and a fixed polynomial in fp32 (IEEE add / multiply / divide only, no contraction: the same bits on the
// host and on the device), accurate to ~1.5 ulp -- as close to the reference's hardware functions as those are to each other.
// (round 3: with these the raw traced images are compared bit for bit; until then a word was allowed to differ by one code.)
RT_MATH float exp2Contract(float x) {
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
RT_MATH float log2Contract(float x) {
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
#undef RT_MATH

RT_DEV uint32_t ftou(float x) { if (!(x > 0.0f)) return 0u; if (x >= 4294967296.0f) return 0xFFFFFFFFu; return (uint32_t)x; }

// Row-vector matrix, row-major: v' = v * M.
struct M4 { float m[4][4]; };
// cbuffer images (rtggx.h): logical M[i][j] = f[j*4+i]
RT_HD M4 cbLoad4x4(const float* f) { M4 r; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = f[j * 4 + i]; return r; }
RT_HD M4 cbLoad4x3(const float* f) {
  M4 r;
  for (int i = 0; i < 4; ++i) { for (int j = 0; j < 3; ++j) r.m[i][j] = f[j * 4 + i]; r.m[i][3] = i == 3 ? 1.0f : 0.0f; }
  return r;
}
RT_HD M4 cbLoad3x3(const float* f) {
  M4 r;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = (i < 3 && j < 3) ? f[j * 4 + i] : (i == j ? 1.0f : 0.0f);
  return r;
}
RT_HD f4 mulPoint(f3 p, const M4& M) {
  f4 r;
  r.x = ((p.x * M.m[0][0] + p.y * M.m[1][0]) + p.z * M.m[2][0]) + M.m[3][0];
  r.y = ((p.x * M.m[0][1] + p.y * M.m[1][1]) + p.z * M.m[2][1]) + M.m[3][1];
  r.z = ((p.x * M.m[0][2] + p.y * M.m[1][2]) + p.z * M.m[2][2]) + M.m[3][2];
  r.w = ((p.x * M.m[0][3] + p.y * M.m[1][3]) + p.z * M.m[2][3]) + M.m[3][3];
  return r;
}
RT_HD f4 mulVec4(f4 p, const M4& M) {
  f4 r;
  r.x = ((p.x * M.m[0][0] + p.y * M.m[1][0]) + p.z * M.m[2][0]) + p.w * M.m[3][0];
  r.y = ((p.x * M.m[0][1] + p.y * M.m[1][1]) + p.z * M.m[2][1]) + p.w * M.m[3][1];
  r.z = ((p.x * M.m[0][2] + p.y * M.m[1][2]) + p.z * M.m[2][2]) + p.w * M.m[3][2];
  r.w = ((p.x * M.m[0][3] + p.y * M.m[1][3]) + p.z * M.m[2][3]) + p.w * M.m[3][3];
  return r;
}
RT_HD f3 mulDir(f3 n, const M4& M) {
  return mk3((n.x * M.m[0][0] + n.y * M.m[1][0]) + n.z * M.m[2][0],
             (n.x * M.m[0][1] + n.y * M.m[1][1]) + n.z * M.m[2][1],
             (n.x * M.m[0][2] + n.y * M.m[1][2]) + n.z * M.m[2][2]);
}

// ---- texel formats (D3D conversion rules; choices listed in DESIGN.md) ---------------------------
RT_DEV uint32_t f2u(float f) { return __float_as_uint(f); }
RT_DEV float u2f(uint32_t u) { return __uint_as_float(u); }

RT_DEV uint32_t f32ToF16(float f) {   // round to nearest even, overflow to inf: v_cvt_f16_f32
  const _Float16 h = (_Float16)f;
  unsigned short u; __builtin_memcpy(&u, &h, 2);
  return (uint32_t)u;
}
RT_DEV float f16ToF32(uint32_t h) {
  const unsigned short u = (unsigned short)h;
  _Float16 x; __builtin_memcpy(&x, &u, 2);
  return (float)x;
}

// unsigned small float of R11G11B10_FLOAT: 5-bit exponent, mbits mantissa; RTNE; negatives -> 0;
// finite overflow saturates to the largest finite value.
RT_DEV uint32_t f32ToUfloat(float f, int mbits) {
  const uint32_t u = f2u(f);
  const uint32_t expMax = 31u << mbits;
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return expMax | 1u;
  if (u & 0x80000000u) return 0u;
  if (u == 0x7F800000u) return expMax;
  const uint32_t maxBits = 0x47000000u | (((1u << mbits) - 1u) << (23 - mbits));
  if (u > maxBits) return expMax - 1u;
  const int shift = 23 - mbits;
  if (u < 0x38800000u) {
    const uint32_t e = u >> 23;
    if (e < 127u - 15u - (uint32_t)mbits - 1u) return 0u;
    const uint32_t mant = (u & 0x7FFFFFu) | 0x800000u;
    const uint32_t sh = (uint32_t)shift + (113u - e);
    if (sh > 31u) return 0u;
    const uint32_t q = mant >> sh, rem = mant & ((1u << sh) - 1u), half = 1u << (sh - 1);
    uint32_t r = q;
    if (rem > half || (rem == half && (q & 1u))) ++r;
    return r;
  }
  uint32_t a = u - 0x38000000u;
  a += ((1u << (shift - 1)) - 1u) + ((a >> shift) & 1u);
  return a >> shift;
}
RT_DEV float ufloatToF32(uint32_t v, int mbits) {
  const uint32_t e = v >> mbits, m = v & ((1u << mbits) - 1u);
  if (e == 0) return (float)m * u2f((127u - 14u - (uint32_t)mbits) << 23);
  if (e == 31) return u2f(0x7F800000u | (m << (23 - mbits)));
  return u2f(((e + 112u) << 23) | (m << (23 - mbits)));
}
RT_DEV uint32_t packR11G11B10F(f3 c) { return f32ToUfloat(c.x, 6) | (f32ToUfloat(c.y, 6) << 11) | (f32ToUfloat(c.z, 5) << 22); }
RT_DEV f3 unpackR11G11B10F(uint32_t p) { return mk3(ufloatToF32(p & 0x7FFu, 6), ufloatToF32((p >> 11) & 0x7FFu, 6), ufloatToF32(p >> 22, 5)); }

RT_DEV uint32_t f32ToUnorm(float x, uint32_t maxv) {
  if (!(x > 0.0f)) return 0u;
  if (x >= 1.0f) return maxv;
  return (uint32_t)(x * (float)maxv + 0.5f);
}
RT_DEV uint32_t packR10G10B10A2(float x, float y, float z, float w) {
  return f32ToUnorm(x, 1023) | (f32ToUnorm(y, 1023) << 10) | (f32ToUnorm(z, 1023) << 20) | (f32ToUnorm(w, 3) << 30);
}
RT_DEV uint32_t packR8G8(float x, float y) { return f32ToUnorm(x, 255) | (f32ToUnorm(y, 255) << 8); }
// velocity store: the sign of a zero is canonicalised to +0 (not observable by any consumer)
RT_DEV uint32_t packR16G16F(float x, float y) {
  uint32_t hx = f32ToF16(x), hy = f32ToF16(y);
  if ((hx & 0x7FFFu) == 0) hx = 0;
  if ((hy & 0x7FFFu) == 0) hy = 0;
  return hx | (hy << 16);
}
RT_DEV uint2 packRGBA16F(float r, float g, float b, float a) {
  return make_uint2(f32ToF16(r) | (f32ToF16(g) << 16), f32ToF16(b) | (f32ToF16(a) << 16));
}
RT_DEV f4 unpackRGBA16F(uint2 p) {
  f4 r; r.x = f16ToF32(p.x & 0xFFFFu); r.y = f16ToF32(p.x >> 16); r.z = f16ToF32(p.y & 0xFFFFu); r.w = f16ToF32(p.y >> 16); return r;
}
RT_DEV uint32_t packRGBA8(float r, float g, float b, float a) {
  return f32ToUnorm(r, 255) | (f32ToUnorm(g, 255) << 8) | (f32ToUnorm(b, 255) << 16) | (f32ToUnorm(a, 255) << 24);
}

// ---- BVH layout (DESIGN.md "BVH layout") ----------------------------------------------------------
struct BvhNode {           // 64 B, internal nodes only; node 0 is the root
  float lmin[3], lmax[3];
  float rmin[3], rmax[3];
  int32_t left, right;     // >= 0 internal node, < 0 leaf: ~ref = slot in the triangle array
  int32_t pad[2];
};
// The 4-wide node the trace kernel walks: binary node i with the internal children of largest surface area folded in (their children
// become its own; lbvh.hip "the 4-wide collapse").  128 B = one L2/L1 cache line per traversal step.  Stored at index i of a sparse
// array (the slots of binary nodes that were folded into another stay unused).
#define RT_BVH4_EMPTY 0x7FFFFFFF
struct Bvh4Node {
  float minx[4], miny[4], minz[4], maxx[4], maxy[4], maxz[4];
  int32_t ref[4];          // >= 0: 4-wide node (= binary node index), < 0: leaf ~slot, RT_BVH4_EMPTY: unused entry
  int32_t pad[4];
};
// The top of each tree a second time, for the trace kernel's LDS (trace.hip): the first RT_TOP_SLOT0 / RT_TOP_SLOT1 4-wide nodes of
// mesh 0 / mesh 1 in breadth-first order, as Bvh4Node records whose references to nodes INSIDE the table read RT_TOP_FLAG | rank
// (rank = position in the table); references to nodes outside it, to leaves and empty entries are unchanged.
#define RT_TOP_FLAG 0x40000000
#ifndef RT_TOP_SLOT0
#define RT_TOP_SLOT0 16
#endif
#ifndef RT_TOP_SLOT1
#define RT_TOP_SLOT1 96
#endif
#define RT_TOP_NODES (RT_TOP_SLOT0 + RT_TOP_SLOT1)
struct BvhTri {            // 64 B (same record size as a node: one cooperative 64-byte gather serves both)
  float v0[3], v1[3], v2[3];
  uint32_t pad0[3];
  uint32_t prim;           // word 12: same position as a node's left-child word, so one 4th quarter serves both record kinds
  uint32_t pad1[3];
};

}  // namespace rt
