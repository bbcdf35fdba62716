// Denoiser chain for gfx950: Denoiser::Denoise / ToneMap (RayTracedGGX/Content/Denoiser.cpp:66-103,
// 361-478) and their shaders:
//   CSSpatial_H_Refl.hlsl:15-50   CSSpatial_V_Refl.hlsl:16-59   (SpatialFilter.hlsli:57-67)
//   CSSpatial_H_Diff.hlsl:15-48   CSSpatial_V_Diff.hlsl:17-59   (SpatialFilter.hlsli:69-83)
//   CSTemporalSS.hlsl:254-336 (_DENOISE_, _ALPHA_AS_ID_, _VARIANCE_AABB_, _USE_YCOCG_; HALF = fp32)
//   PSToneMap.hlsl:13-41
// Ping-pong as in SURVEY.md Appendix C: scratch = TSS[parity], history = TSS[!parity].
//
// Spatial passes, two variants as in the reference (Denoiser.cpp:375-405, key [V]):
//   shared memory (use_shared_mem = 1): a workgroup stages its pixels plus the 16-texel aprons ONCE in LDS, already
//     unpacked into the eight floats the 33-tap loop consumes (normal*2-1, depth, roughness, tone-mapped colour;
//     structure-of-arrays, so that the lanes of a wave read consecutive words: conflict-free).  H pass: 64x4 pixels
//     from a 96x4 tile (12 KB); V pass: 16x16 pixels from a 16x48 tile (24 KB).
//   direct access (0): every tap fetches and unpacks its texel from global memory (L1/L2 hits).
// Both evaluate the same weights: pow(x, 512) / pow(x, 32) as 9 / 5 squarings, and the Gaussian and depth
// exponentials merged into one exp2 -- ~35 VALU instructions per tap instead of ~400 with libm calls.  The passes
// are VALU-bound (33 taps per covered pixel); their HBM traffic (22-30 B/pixel/pass) is ~5% of the
// time.  Out-of-range texels are the zeros D3D returns.
#include <cstring>
#include <type_traits>
#include <hip/hip_ext.h>
#include "rtggx_context.h"

namespace rt {

#define RT_RADIUS 16
// Block of the vertical passes.  Measured on the 1080p bunny frame (reflection V pass): 32x32 (64 KB tile) 64 us,
// 16x32 42 us, 16x16 (24 KB) 38 us -- residency beats the larger apron share of small tiles.
#ifndef RT_VBW
#define RT_VBW 16
#define RT_VBH 16
#endif

// The denoiser's outputs are floating-point images judged to 1e-3 relative L2 (DESIGN.md "parity bar"), and with both
// streams busy the frame is bound by VALU issue: in the spatial filters and the tone map, IEEE divisions (10-12
// instructions each) become multiplications by a constant or by v_rcp_f32 (1 ulp).  Not in the temporal pass: see tssTM.
RT_DEV float rcpFast(float x) { return __builtin_amdgcn_rcpf(x); }

// nx, ny, nz: the 10-bit UNORM normal as the INTEGERS 2 k - 1023 (in floats): the reference's value k * 2 / 1023 - 1 is that / 1023, and the
// filters' dot products are then exact in fp32 (tapWeight)
struct GTexel { float nx, ny, nz, nw, rough, metal, depth; };

RT_DEV GTexel loadG(const uint32_t* __restrict__ normal, const uint16_t* __restrict__ roughMetal, const uint32_t* __restrict__ depth32,
                    int x, int y, int W, int H) {
  GTexel g;
  if (x < 0 || y < 0 || x >= W || y >= H) { g.nx = g.ny = g.nz = -1023.0f; g.nw = 0.0f; g.rough = 0.0f; g.metal = 0.0f; g.depth = 0.0f; return g; }      // (zeros: the normal reads -1)
  const size_t i = (size_t)y * W + x;
  const uint32_t n = normal[i];
  g.nx = (float)(2 * (int)(n & 1023u) - 1023);
  g.ny = (float)(2 * (int)((n >> 10) & 1023u) - 1023);
  g.nz = (float)(2 * (int)((n >> 20) & 1023u) - 1023);
  g.nw = (float)(n >> 30) * (1.0f / 3.0f);
  const uint32_t rm = roughMetal[i];
  g.rough = (float)(rm & 0xFFu) * (1.0f / 255.0f); g.metal = (rm >> 8) == 255u ? 1.0f : (float)(rm >> 8) * (1.0f / 255.0f);
  g.depth = (float)depth32[i] * (1.0f / 16777215.0f);
  return g;
}
RT_DEV f3 TM3(f3 c) { const float r = rcpFast(1.0f + ((c.x * 0.25f + c.y * 0.5f) + c.z * 0.25f)); return mk3(c.x * r, c.y * r, c.z * r); }     // FilterCommon.hlsli:14-19
RT_DEV f3 ITM3(f3 c) { const float r = rcpFast(1.0f - ((c.x * 0.25f + c.y * 0.5f) + c.z * 0.25f)); return mk3(c.x * r, c.y * r, c.z * r); }    // :24-27
struct Targets {
  const uint32_t* normal; const uint16_t* roughMetal; const uint32_t* depth32; const uint32_t* velocity;
  const uint32_t* rtRefl; const uint32_t* rtDiff;
  uint2* scratch; const uint2* history; uint2* fltRfl; uint2* fltDff; uint32_t* backbuffer;
  int W, H, rowBegin, rowEnd;
  // Strips of a multi-GPU frame: rows [histLo, histHi) of `history` are valid on this rank (its own rows + the apron its
  // neighbours delivered).  A reprojection that reads beyond them records by how many rows (histReach; null on whole frames).
  int histLo, histHi; uint32_t* histReach;
  // ... and beyond them the tap reads the OWNER's image directly (round 4): peerHist[r] = rank r's TemporalSSOut[!parity] as mapped into this
  // process (rtggx_set_history_peers), peerBounds[r] .. peerBounds[r + 1] the rows rank r owns.  Null: no peers (the tap reads this rank's
  // own buffer whatever it holds there -- rounds 2-3's "reported, not prevented").
  const uint2* const* peerHist; const uint32_t* peerBounds; int peerWorld;
  int outBegin, outEnd;      // rows of the back buffer (the strip itself): the fused temporal + tone-map kernel
  // one word per 16x16 tile, counted from row tileRow0, tilesX to a row: 0 = the visibility pass drew nothing there, no pixel of it has a surface
  // (rtggx_context.h visDirtyBuf; all ones where that is not known)
  const uint32_t* tileWords; int tilesX, tileRow0;
};
#define RT_SGPR(v) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(v)))      // a workgroup-uniform value the compiler may have computed in vector registers
// Four of those words OR-ed together: scalar loads, one wait (the indices are uniform over the workgroup).
RT_DEV uint32_t tileWordsOr(const uint32_t* words, uint32_t i0, uint32_t i1, uint32_t i2, uint32_t i3) {
  uint32_t a, b, c, d;
  asm volatile("s_load_dword %0, %4, %5\n\ts_load_dword %1, %4, %6\n\ts_load_dword %2, %4, %7\n\ts_load_dword %3, %4, %8\n\ts_waitcnt lgkmcnt(0)"
               : "=&s"(a), "=&s"(b), "=&s"(c), "=&s"(d) : "s"(words), "s"(RT_SGPR(i0 * 4u)), "s"(RT_SGPR(i1 * 4u)), "s"(RT_SGPR(i2 * 4u)), "s"(RT_SGPR(i3 * 4u)) : "memory");
  return a | b | c | d;
}

RT_DEV uint32_t tileWordsOr6(const uint32_t* words, uint32_t i0, uint32_t i1, uint32_t i2, uint32_t i3, uint32_t i4, uint32_t i5) {
  uint32_t a, b, c, d, e, f;
  asm volatile("s_load_dword %0, %6, %7\n\ts_load_dword %1, %6, %8\n\ts_load_dword %2, %6, %9\n\ts_load_dword %3, %6, %10\n\ts_load_dword %4, %6, %11\n\ts_load_dword %5, %6, %12\n\ts_waitcnt lgkmcnt(0)"
               : "=&s"(a), "=&s"(b), "=&s"(c), "=&s"(d), "=&s"(e), "=&s"(f)
               : "s"(words), "s"(RT_SGPR(i0 * 4u)), "s"(RT_SGPR(i1 * 4u)), "s"(RT_SGPR(i2 * 4u)), "s"(RT_SGPR(i3 * 4u)), "s"(RT_SGPR(i4 * 4u)), "s"(RT_SGPR(i5 * 4u)) : "memory");
  return a | b | c | d | e | f;
}

#define RT_LOG2E 1.44269504088896341f

// Weight of one tap (SpatialFilter.hlsli:57-75, FilterCommon.hlsli:34-42,59-71) given the centre's constants.
//   reflection: [nw > 0] * Gaussian(i, blurRadius) * pow(max(N.Nc, 0), 512) * exp(-|dc - d| dc 4) * (1 - smoothstep(0, .5, |r - rc|))
//   diffuse:    pow(max(N.Nc, 0), 32) * exp(-|dc - d| dc 4)
// The diffuse loops SKIP texels that do not take part (nw = 0 or metal = 1; CSSpatial_H_Diff.hlsl:35): those are staged
// with a zero normal and colour.  The reflection weight instead MULTIPLIES by the nw flag (SpatialFilter.hlsli:60):
// an out-of-range texel reads as zeros, i.e. normal (-1,-1,-1), and for a centre normal with nx+ny+nz < -1.19 the
// 512th power overflows, 0 x inf = NaN, and the pixel (then its column) turns NaN -- in the reference, hence here:
// reflection texels are staged as they are, with the flag in the sign bit of the roughness word.
struct Centre { float nx, ny, nz, depth, rough, gaussK /* -0.5 log2(e) / sigma^2 */, depthK /* dc 4 log2(e) */; };
// The normal weight pow(max(dot(N, Nc), 0), 512 | 32) (SpatialFilter.hlsli:62,73).  x = dot(N, Nc) sits next to 1 and the power multiplies
// its relative error by 512: one fp32 rounding of the dot product (6e-8) is 3e-5 in the weight, and the temporal pass turns a 1e-5
// difference of the filtered image into 1e-3 of its result.  HLSL leaves the order of a dp3 and the last bits of
// pow (= exp2(y log2 x) on 1-ulp functions) open, so two faithful implementations differ by that much; rounds 1-3 passed the parity
// check by sharing one arbitrary choice (a fused dot product and nine squarings, 423 ulps from the true power) with the oracle.
// Round 4 evaluates the TRUE value instead, to a few ulps, and the oracle holds its own evaluations against it (exact in double; plain
// fp32 libm):
//   * the normals are 10-bit UNORM codes k: N = (2 k - 1023) / 1023.  With the integers 2 k - 1023 the dot product I is an integer below
//     2^24: three fp32 operations, no rounding at all.  x - 1 = d = (I - 1023^2) / 1023^2: exact numerator, ONE rounding, relative to d;
//   * log2(1 + d) = v_log_f32(fl(1 + d)) + (what that rounding dropped) log2(e);
//   * the power, the Gaussian and the depth term share one v_exp_f32: 2^(n log2 x + e).  A power that overflows (x >= 2^(1/4): only a
//     tap outside the frame, whose normal reads (-1, -1, -1)) stays +inf whatever e is -- the reference's 0 x inf = NaN for such taps
//     (SpatialFilter.hlsli:60) appears in the same pixels.
// tools/microbench/pow512.hip, profiles/r04_a_pow512.txt: within 5 ulps of the true power where the weight is above 0.01.
#define RT_NORMAL_SCALE_SQ 1046529.0f      // 1023^2
RT_DEV float log2OfDot(float I) {          // log2(max(I / 1023^2, 0))
  const float d = fmaxf((I - RT_NORMAL_SCALE_SQ) * (1.0f / RT_NORMAL_SCALE_SQ), -1.0f);
  const float xr = 1.0f + d;
  const float dropped = (1.0f - xr) + d;
  return __builtin_fmaf(dropped, RT_LOG2E, __builtin_amdgcn_logf(xr));      // (x = 0: -inf, the weight 0)
}
// UNIFORM (round 4; the tiled kernels decide it per workgroup while they stage their tile): 1 -- every texel of the tile that has a
// surface has the centre's roughness, so the roughness weight is 1 for every tap that counts (a tap without a surface is multiplied by
// its zero flag whatever that weight would be) and its seven operations are left out; 2 -- and every texel has a surface: the flag is
// left out as well.  Same bits (x 1.0 is exact); most tiles inside the model or the ground are such tiles.
template <bool DIFFUSE, int UNIFORM = 0>
RT_DEV float tapWeight(const Centre& c, int i, float nx, float ny, float nz, float depth, float rough) {
#pragma clang fp contract(fast)
  const float I = __builtin_fmaf(c.nz, nz, __builtin_fmaf(c.ny, ny, c.nx * nx));      // exact: integers below 2^24
  const float lg = log2OfDot(I);
  const float dd = fabsf(c.depth - depth) * c.depthK;
  if (DIFFUSE) return __builtin_amdgcn_exp2f(__builtin_fmaf(32.0f, lg, -dd));
  const float pw = 512.0f * lg;
  const float e = __builtin_fmaf(c.gaussK, (float)(i * i), -dd);
  const float w = __builtin_amdgcn_exp2f(pw >= 128.0f ? pw : pw + e);
  if (UNIFORM == 2) return w;
  const float flag = (__float_as_uint(rough) >> 31) ? 0.0f : 1.0f;          // sign bit set: norm.w <= 0
  if (UNIFORM == 1) return w * flag;
  const float t = saturatef(fabsf(fabsf(rough) - c.rough) * 2.0f);
  return (w * (1.0f - t * t * (3.0f - 2.0f * t))) * flag;
}
template <bool DIFFUSE>
RT_DEV Centre makeCentre(float nx, float ny, float nz, float depth, float rough, int W, int H) {
  Centre c; c.nx = nx; c.ny = ny; c.nz = nz; c.depth = depth; c.rough = rough;
  const int br = DIFFUSE ? 0 : (int)clampf(0.1f * rough * (float)W, 0.0f, (float)H * 0.05f);   // FilterCommon.hlsli:49-52
  const float sigma = (float)(br + 1) * (1.0f / 3.0f);
  c.gaussK = (-0.5f * RT_LOG2E) * rcpFast(sigma * sigma);
  c.depthK = depth * (4.0f * RT_LOG2E);
  return c;
}
template <int MODE>
RT_DEV void storeFiltered(const Targets& T, size_t pix, float mx, float my, float mz, float wsum) {
  const float rw = rcpFast(wsum);
  f3 mu = mk3(mx * rw, my * rw, mz * rw);
  if (MODE == 0 || MODE == 2) T.scratch[pix] = packRGBA16F(mu.x, mu.y, mu.z, 0.0f);
  if (MODE == 1) { mu = ITM3(mu); const uint2 v = packRGBA16F(mu.x, mu.y, mu.z, 1.0f); if (T.fltRfl) T.fltRfl[pix] = v; T.fltDff[pix] = v; }
  if (MODE == 3) {
    const f4 dest = unpackRGBA16F(T.fltRfl[pix]);
    mu = ITM3(mu);
    T.fltDff[pix] = packRGBA16F(dest.x + mu.x, dest.y + mu.y, dest.z + mu.z, dest.w);
  }
}
// Pixels the pass does not filter (no surface; diffuse: pure metal) -- CSSpatial_V_Refl.hlsl:27-31, CSSpatial_V_Diff.hlsl:28-32.
// FilteredOut1 = FilteredOut + diffuse: the reflection V pass writes its result to both targets, so the diffuse V pass
// only touches the pixels it filters (on the all-metal default scene: none) instead of copying the whole image.
template <int MODE>
RT_DEV void storeSkipped(const Targets& T, size_t pix) {
  if (MODE == 1) { const f3 s = unpackR11G11B10F(T.rtRefl[pix]); const uint2 v = packRGBA16F(s.x, s.y, s.z, 0.0f); if (T.fltRfl) T.fltRfl[pix] = v; T.fltDff[pix] = v; }
}
// Source colour of a tap: the ray-traced result tone-mapped (H passes) or the H pass's scratch (V passes).
template <int MODE>
RT_DEV f3 tapColour(const Targets& T, size_t ti) {
  if (MODE & 1) { const f4 v = unpackRGBA16F(T.scratch[ti]); return mk3(v.x, v.y, v.z); }
  return TM3(unpackR11G11B10F(MODE >= 2 ? T.rtDiff[ti] : T.rtRefl[ti]));
}

// mode 0: H_Refl  1: V_Refl  2: H_Diff  3: V_Diff -- direct-access variant
template <int MODE>
__global__ void __launch_bounds__(256) spatialDirectKernel(Targets T) {
  constexpr bool vertical = (MODE & 1) != 0;
  constexpr bool diffuse = MODE >= 2;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = T.rowBegin + blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= T.W || y >= T.rowEnd) return;
  const size_t pix = (size_t)y * T.W + x;
  const GTexel gc = loadG(T.normal, T.roughMetal, T.depth32, x, y, T.W, T.H);
  if (diffuse ? (gc.nw <= 0.0f || gc.metal >= 1.0f) : (gc.nw <= 0.0f)) { storeSkipped<MODE>(T, pix); return; }
  const Centre c = makeCentre<diffuse>(gc.nx, gc.ny, gc.nz, gc.depth, gc.rough, T.W, T.H);
  float mx = 0.0f, my = 0.0f, mz = 0.0f, wsum = 0.0f;
  for (int i = -RT_RADIUS; i <= RT_RADIUS; ++i) {
    const int tx = vertical ? x : x + i, ty = vertical ? y + i : y;
    const bool inside = tx >= 0 && ty >= 0 && tx < T.W && ty < T.H;
    const GTexel g = loadG(T.normal, T.roughMetal, T.depth32, tx, ty, T.W, T.H);     // zeros outside: normal -1, flag 0
    if (diffuse && (g.nw <= 0.0f || g.metal >= 1.0f)) continue;
    const f3 src = inside ? tapColour<MODE>(T, (size_t)ty * T.W + tx) : mk3(0.0f, 0.0f, 0.0f);
    const float w = tapWeight<diffuse>(c, i, g.nx, g.ny, g.nz, g.depth, g.nw > 0.0f ? g.rough : __uint_as_float(__float_as_uint(g.rough) | 0x80000000u));
    mx = __builtin_fmaf(src.x, w, mx); my = __builtin_fmaf(src.y, w, my); mz = __builtin_fmaf(src.z, w, mz);
    wsum += w;
  }
  storeFiltered<MODE>(T, pix, mx, my, mz, wsum);
}

// Shared-memory variant.  Block geometry: H passes 64x4 pixels (one per thread), tile 96x4; V passes RT_VBW x RT_VBH
// pixels (thread (lx, ly) filters rows ly, ly + 256/RT_VBW, ... of column lx), tile RT_VBW x (RT_VBH + 32).
typedef float V4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void __launch_bounds__(256) spatialTiledKernel(Targets T) {
  constexpr bool vertical = (MODE & 1) != 0;
  constexpr bool diffuse = MODE >= 2;
  constexpr int BW = vertical ? RT_VBW : 64, BH = vertical ? RT_VBH : 4;      // pixels of the block
  constexpr int TW = vertical ? RT_VBW : 96, TH = vertical ? RT_VBH + 2 * RT_RADIUS : 4;      // texels of the tile
  constexpr int PER = vertical ? BW * BH / 256 : 1, ROWSTEP = 256 / BW;           // pixels per thread, their row distance
  constexpr int N = TW * TH;
  // Two 16-byte words per texel, texel after texel: a tap is two ds_read_b128 -- 256 bytes per clock and CU, where the eight ds_read_b32 of a
  // field-major tile get 128 (MI355X_MICROARCH.md, LDS) and, at 16 LDS cycles per tap and wave against ~13 cycles of vector issue, set the
  // pace of the tap loop.  Consecutive lanes read consecutive texels in both passes: conflict-free.
  __shared__ V4 smA[N], smB[N];                                       // nx ny nz depth | rough r g b
  __shared__ uint32_t roughLo, roughHi, unflagged;                    // over the tile: least / greatest roughness of the texels with a surface, texels without one
  if (threadIdx.x == 0) { roughLo = 0xFFFFFFFFu; roughHi = 0u; unflagged = 0u; }
  const int bx0 = blockIdx.x * BW, by0 = T.rowBegin + blockIdx.y * BH;
  const int lx = threadIdx.x % BW, ly = threadIdx.x / BW;
  const int x = bx0 + lx;

  // Nothing drawn in the tiles the block lies in (three quarters of the bunny frame): no pixel has a surface, none is filtered -- known
  // from four scalar loads instead of a read of the normals (whose latency was most of such a workgroup's life)
  static_assert(RT_VBW <= 32 && RT_VBH <= 32, "the V block spans at most two tiles across and three down");
  bool nothingDrawn;
  { const uint32_t txLast = (uint32_t)T.tilesX - 1u, tx0 = (uint32_t)bx0 >> 4, tx1 = min(tx0 + 1u, txLast);
    const uint32_t ty0 = (uint32_t)(by0 - T.tileRow0) >> 4, ty1 = (uint32_t)(min(by0 + BH, T.rowEnd) - 1 - T.tileRow0) >> 4;
    const uint32_t r0 = ty0 * (uint32_t)T.tilesX, r1 = ty1 * (uint32_t)T.tilesX, rm = ((ty0 + ty1) >> 1) * (uint32_t)T.tilesX;
    nothingDrawn = vertical ? tileWordsOr6(T.tileWords, r0 + tx0, r0 + tx1, rm + tx0, rm + tx1, r1 + tx0, r1 + tx1) == 0u
                            : tileWordsOr(T.tileWords, r0 + tx0, r0 + tx1, r0 + min(tx0 + 2u, txLast), r0 + min(tx0 + 3u, txLast)) == 0u; }
  if (nothingDrawn) {
    if (MODE == 1) {
#pragma unroll
      for (int k = 0; k < PER; ++k) { const int y = by0 + ly + k * ROWSTEP; if (x < T.W && y < T.rowEnd) storeSkipped<MODE>(T, (size_t)y * T.W + x); }
    }
    return;
  }
  // which of my pixels are filtered at all; the others get their pass-through value now
  bool todo[PER]; bool any = false;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int y = by0 + ly + k * ROWSTEP;
    todo[k] = false;
    if (x < T.W && y < T.rowEnd) {
      const size_t pix = (size_t)y * T.W + x;
      const bool skip = (T.normal[pix] >> 30) == 0u || (diffuse && (T.roughMetal[pix] >> 8) == 255u);
      if (skip) storeSkipped<MODE>(T, pix); else { todo[k] = true; any = true; }
    }
  }
  if (!__syncthreads_or(any ? 1 : 0)) return;

  // stage the tile
  const int ox = vertical ? bx0 : bx0 - RT_RADIUS, oy = vertical ? by0 - RT_RADIUS : by0;
  uint32_t myLo = 0xFFFFFFFFu, myHi = 0u, myUnflagged = 0u;
  for (int t = threadIdx.x; t < N; t += 256) {
    const int tx = ox + t % TW, ty = oy + t / TW;
    float v[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    const bool inside = tx >= 0 && ty >= 0 && tx < T.W && ty < T.H;
    const GTexel g = loadG(T.normal, T.roughMetal, T.depth32, tx, ty, T.W, T.H);       // zeros outside
    v[3] = g.depth; v[4] = g.rough;
    if (diffuse) {
      if (g.nw > 0.0f && g.metal < 1.0f) {
        const f3 src = tapColour<MODE>(T, (size_t)ty * T.W + tx);
        v[0] = g.nx; v[1] = g.ny; v[2] = g.nz; v[5] = src.x; v[6] = src.y; v[7] = src.z;
      }
    } else {
      v[0] = g.nx; v[1] = g.ny; v[2] = g.nz;
      if (g.nw <= 0.0f) { v[4] = __uint_as_float(__float_as_uint(g.rough) | 0x80000000u); myUnflagged = 1u; }
      else { myLo = min(myLo, __float_as_uint(g.rough)); myHi = max(myHi, __float_as_uint(g.rough)); }
      if (inside) { const f3 src = tapColour<MODE>(T, (size_t)ty * T.W + tx); v[5] = src.x; v[6] = src.y; v[7] = src.z; }
    }
    smA[t] = V4{v[0], v[1], v[2], v[3]}; smB[t] = V4{v[4], v[5], v[6], v[7]};
  }
  if (!diffuse) {      // one LDS atomic per wave and fact
    for (int o = 32; o > 0; o >>= 1) { myLo = min(myLo, (uint32_t)__shfl_down((int)myLo, o)); myHi = max(myHi, (uint32_t)__shfl_down((int)myHi, o)); }
    const unsigned long long anyUnflagged = __ballot(myUnflagged != 0u);
    if ((threadIdx.x & 63) == 0) { atomicMin(&roughLo, myLo); atomicMax(&roughHi, myHi); if (anyUnflagged) atomicOr(&unflagged, 1u); }
  }
  __syncthreads();
  const int uniform = diffuse ? 0 : (roughLo != roughHi ? 0 : unflagged ? 1 : 2);      // (the same for the whole workgroup: see tapWeight)

  auto filter = [&](auto uniformTag) {
    constexpr int UNIFORM = decltype(uniformTag)::value;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      if (!todo[k]) continue;
      const int y = by0 + ly + k * ROWSTEP;
      const int ci = vertical ? (ly + k * ROWSTEP + RT_RADIUS) * TW + lx : ly * TW + lx + RT_RADIUS;
      const V4 ca = smA[ci];
      const Centre c = makeCentre<diffuse>(ca.x, ca.y, ca.z, ca.w, smB[ci].x, T.W, T.H);
      float mx = 0.0f, my = 0.0f, mz = 0.0f, wsum = 0.0f;
      // one LDS address per word (the first tap's), held in a register: the 33 taps are then immediate offsets of the
      // ds_read instructions instead of an address computation each
      typedef __attribute__((address_space(3))) const V4 LdsV4;
      LdsV4* pa = (LdsV4*)&smA[ci - RT_RADIUS * (vertical ? TW : 1)]; LdsV4* pb = (LdsV4*)&smB[ci - RT_RADIUS * (vertical ? TW : 1)];
      asm volatile("" : "+v"(pa)); asm volatile("" : "+v"(pb));
#pragma unroll
      for (int i = -RT_RADIUS; i <= RT_RADIUS; ++i) {
        const int ti = (i + RT_RADIUS) * (vertical ? TW : 1);
        const V4 a = pa[ti], b = pb[ti];
        const float w = tapWeight<diffuse, UNIFORM>(c, i, a.x, a.y, a.z, a.w, b.x);
        mx = __builtin_fmaf(b.y, w, mx); my = __builtin_fmaf(b.z, w, my); mz = __builtin_fmaf(b.w, w, mz);
        wsum += w;
      }
      storeFiltered<MODE>(T, (size_t)y * T.W + x, mx, my, mz, wsum);
    }
  };
  if (uniform == 2) filter(std::integral_constant<int, 2>{});
  else if (uniform == 1) filter(std::integral_constant<int, 1>{});
  else filter(std::integral_constant<int, 0>{});
}

// ---- CSTemporalSS.hlsl --------------------------------------------------------------------------------
RT_DEV f3 rgbToYCoCg(f3 c) {   // :78-85
  return mk3((c.x * 1.0f + c.y * 2.0f) + c.z * 1.0f, (c.x * 2.0f + c.y * 0.0f) + c.z * -2.0f, (c.x * -1.0f + c.y * 2.0f) + c.z * -1.0f);
}
RT_DEV f3 yCoCgToRGB(f3 c) {   // :90-101
  const float y = c.x * 0.25f, co = c.y * 0.25f, cg = c.z * 0.25f;
  return mk3(y + co - cg, y + cg, y - co - cg);
}
// The temporal pass keeps correctly rounded division and square root: its neighbourhood variance m2/9 - mu^2 is pure
// rounding noise in flat regions, sqrt turns that noise into the clamp window (times gamma <= 32), and only identical
// arithmetic on both sides keeps the window -- and with it the clamped history -- comparable with the oracle's.  They are
// the short exact sequences of rtggx_device.h (divBy9, div3Shared, sqrtRN), not the compiler's full-range expansions.
// The smooth terms around it (reprojection uv, blend factors, the inverse tone map) do use v_rcp_f32.
RT_DEV f3 tssTM(f3 hdr) {   // :106-114; three IEEE quotients that share the refinement of 1/d (rtggx_device.h)
  const f3 c = rgbToYCoCg(hdr); const float d = 4.0f + c.x;
  f3 q; div3Shared(c.x, c.y, c.z, d, q.x, q.y, q.z); return q;
}
RT_DEV f3 tssITM(f3 col) { const float k = 4.0f * rcpFast(1.0f - col.x); return yCoCgToRGB(mk3(col.x * k, col.y * k, col.z * k)); }   // :119-128 (smooth: v_rcp)
RT_DEV f3 tssTMSmooth(f3 hdr) { const f3 c = rgbToYCoCg(hdr); const float r = rcpFast(4.0f + c.x); return mk3(c.x * r, c.y * r, c.z * r); }   // for the reprojected history, which does not enter the variance
RT_DEV f2 loadVel(const uint32_t* __restrict__ vel, int x, int y, int W, int H) {
  f2 v; v.x = 0.0f; v.y = 0.0f;
  if (x < 0 || y < 0 || x >= W || y >= H) return v;
  const uint32_t p = vel[(size_t)y * W + x];
  v.x = f16ToF32(p & 0xFFFFu); v.y = f16ToF32(p >> 16);
  return v;
}
RT_DEV f4 loadRGBA16(const uint2* __restrict__ b, int x, int y, int W, int H) {
  if (x < 0 || y < 0 || x >= W || y >= H) { f4 z; z.x = z.y = z.z = z.w = 0.0f; return z; }
  return unpackRGBA16F(b[(size_t)y * W + x]);
}

// The 3x3 neighbourhood is read from an LDS tile that holds tssTM(FilteredOut1) (alpha kept) of the workgroup's pixels and a one-texel
// apron: each texel is unpacked and tone-mapped (three divisions) once instead of nine times.  ONE pixel per thread: round 3 built
// workgroups of 64 x 8 and 64 x 16 pixels with two and four pixels per thread (1.29 x / 1.16 x the pixels loaded instead of 1.55 x) and
// lost -- bunny 1080p 0.1818 -> 0.1944 -> 0.1970 ms -- because the pixels with a surface take the long path (four 8-byte history taps, the
// clamp window), and a thread that walks several of them one after the other serialises their round trips.
RT_DEV const uint2* peerHistoryRow(const Targets& T, int iy) {
  int r = 0;
  while (r + 1 < T.peerWorld && (int)T.peerBounds[r + 1] <= iy) ++r;
  return T.peerHist[r] + (size_t)iy * T.W;
}
// One pixel of the temporal pass: (x, y) of the frame = (lx, ly) of the workgroup's LDS tiles (tone-mapped FilteredOut1 + alpha, the
// velocity texels and their squared lengths; 66 texels per row, a one-texel apron all round).  Returns TemporalSSOut's packed texel.
template <int ROWS, int PITCH>
RT_DEV uint2 temporalPixel(const Targets& T, const float4 (&tile)[ROWS][PITCH], const uint32_t (&velRaw)[ROWS][PITCH], const float (&velSq)[ROWS][PITCH], int x, int y, int lx, int ly) {
  const int W = T.W, H = T.H;
  const float Wf = (float)W, Hf = (float)H;
  const float uvx = ((float)x + 0.5f) * rcpFast(Wf), uvy = ((float)y + 0.5f) * rcpFast(Hf);
  const float4 cur = tile[ly][lx];
  const f3 currentTM = mk3(cur.x, cur.y, cur.z);
  f4 current; current.x = current.y = current.z = 0.0f; current.w = cur.w;     // only the alpha of the raw value is used below
  // VelocityMax :133-161
  const int ox[8] = {-1, 1, 0, 0, -1, 1, 1, -1}, oy[8] = {0, 0, -1, 1, -1, -1, 1, 1};   // g_texOffsets :48-52
  f2 vmax;
  {
    float speedSq = velSq[ly][lx];
    uint32_t raw = velRaw[ly][lx];
    for (int i = 4; i < 8; ++i) {
      const float sq = velSq[ly + oy[i]][lx + ox[i]];
      const uint32_t r = velRaw[ly + oy[i]][lx + ox[i]];
      if (sq > speedSq) { raw = r; speedSq = sq; }
    }
    vmax.x = f16ToF32(raw & 0xFFFFu); vmax.y = f16ToF32(raw >> 16);
  }
  // The alpha of the 3x3 neighbourhood, filtered like the colour below (fl[3] of NeighborMinMax).  Zero means there is no
  // surface anywhere in the neighbourhood (background: FilteredOut1 carries alpha 0 there), and then blend = 1 at :325:
  // the history does not contribute to the colour.  Such pixels -- most of the frame -- take the short path at the end
  // of this kernel, which leaves out what only feeds the history term: the history colour, the chroma variances, the
  // clamp window.  It computes a + 1 (b - a) as b, so it matches the long path to rounding, not to the bit.
  float alphaFiltered = cur.w;
  for (int i = 0; i < 8; ++i) alphaFiltered = __builtin_fmaf(tile[ly + oy[i]][lx + ox[i]].w, i < 4 ? 0.5f : 0.25f, alphaFiltered);
  alphaFiltered /= 4.0f;
  const bool plain = !(alphaFiltered > 0.0f);
  // history: bilinear, clamped addressing :259-260
  f4 history;
  {
#pragma clang fp contract(fast)       // smooth terms (reprojection, bilinear weights): fused multiply-adds are welcome
    const float sx = (uvx - vmax.x) * Wf - 0.5f, sy = (uvy - vmax.y) * Hf - 0.5f;
    const float x0 = floorf(sx), y0 = floorf(sy);
    const float fx = sx - x0, fy = sy - y0;
    const int ix0 = x0 < 0.0f ? 0 : (x0 > (float)(W - 1) ? W - 1 : (int)x0);
    const int ix1 = x0 + 1.0f < 0.0f ? 0 : (x0 + 1.0f > (float)(W - 1) ? W - 1 : (int)(x0 + 1.0f));
    const int iy0 = y0 < 0.0f ? 0 : (y0 > (float)(H - 1) ? H - 1 : (int)y0);
    const int iy1 = y0 + 1.0f < 0.0f ? 0 : (y0 + 1.0f > (float)(H - 1) ? H - 1 : (int)(y0 + 1.0f));
    const float w00 = (1.0f - fx) * (1.0f - fy), w10 = fx * (1.0f - fy), w01 = (1.0f - fx) * fy, w11 = fx * fy;
    // Strips: rows [histLo, histHi) of `history` are this rank's own and the apron its neighbours delivered; a tap beyond them (a pixel
    // that moved more than the apron's rows in one frame: one sliver triangle per few hundred frames of the turning bunny) is counted
    // (histReach, SURVEY 8e "clamp and report") and read from the image of the rank that OWNS the row -- the reference samples its one
    // history texture anywhere (CSTemporalSS.hlsl:259-265).
    const uint2* row0 = T.history + (size_t)iy0 * W; const uint2* row1 = T.history + (size_t)iy1 * W;
    if (T.histReach != nullptr) {
      const int over = max(T.histLo - iy0, iy1 - (T.histHi - 1));
      if (over > 0) {
        atomicMax(T.histReach, (uint32_t)over);
        if (T.peerHist != nullptr) {
          if (iy0 < T.histLo || iy0 >= T.histHi) row0 = peerHistoryRow(T, iy0);
          if (iy1 < T.histLo || iy1 >= T.histHi) row1 = peerHistoryRow(T, iy1);
        }
      }
    }
    const uint2* h00 = row0 + ix0; const uint2* h10 = row0 + ix1;
    const uint2* h01 = row1 + ix0; const uint2* h11 = row1 + ix1;
    if (plain) {      // only the alpha (history length) is needed
      history.x = history.y = history.z = 0.0f;
      history.w = ((f16ToF32(h00->y >> 16) * w00 + f16ToF32(h10->y >> 16) * w10) + f16ToF32(h01->y >> 16) * w01) + f16ToF32(h11->y >> 16) * w11;
    } else {
      const f4 t00 = unpackRGBA16F(*h00), t10 = unpackRGBA16F(*h10), t01 = unpackRGBA16F(*h01), t11 = unpackRGBA16F(*h11);
      history.x = ((t00.x * w00 + t10.x * w10) + t01.x * w01) + t11.x * w11;
      history.y = ((t00.y * w00 + t10.y * w10) + t01.y * w01) + t11.y * w11;
      history.z = ((t00.z * w00 + t10.z * w10) + t01.z * w01) + t11.z * w11;
      history.w = ((t00.w * w00 + t10.w * w10) + t01.w * w01) + t11.w * w11;
    }
  }
  // :262-281
  float curHistoryBlur = fabsf(vmax.x) * (4.0f * Wf) + fabsf(vmax.y) * (4.0f * Hf);
  float historyBlur = 1.0f - history.w;
  historyBlur = fmaxf(historyBlur, curHistoryBlur);
  history.w = history.w * 15.0f + 1.0f;
  f3 result; float hw;
  if (plain) {
    // NeighborMinMax :166-236, the part that reaches the result when blend = 1: the filtered colour and the luma contrast
    float fl[3] = {currentTM.x, currentTM.y, currentTM.z};
    float mu0 = currentTM.x, m20 = mu0 * mu0;
    for (int i = 0; i < 8; ++i) {
      const float4 t = tile[ly + oy[i]][lx + ox[i]];
      const float wgt = i < 4 ? 0.5f : 0.25f;
      fl[0] = __builtin_fmaf(t.x, wgt, fl[0]); fl[1] = __builtin_fmaf(t.y, wgt, fl[1]); fl[2] = __builtin_fmaf(t.z, wgt, fl[2]);
      mu0 += t.x; m20 += t.x * t.x;
    }
    for (int k = 0; k < 3; ++k) fl[k] /= 4.0f;
    mu0 = divBy9(mu0);
    const float sigma0 = sqrtRN(fabsf(divBy9(m20) - mu0 * mu0));
    const float nmin3 = mu0 - sigma0, nmax3 = mu0 + sigma0;
    {
#pragma clang fp contract(fast)
      curHistoryBlur = saturatef(curHistoryBlur);
      historyBlur = saturatef(historyBlur);
      const float contrast = nmax3 - nmin3;
      float addAlias = historyBlur * 0.5f + 0.25f;
      addAlias = saturatef(addAlias + rcpFast(1.0f + contrast * (32.0f * 4.0f)));
      result = tssITM(mk3(lerpf(fl[0], currentTM.x, addAlias), lerpf(fl[1], currentTM.y, addAlias), lerpf(fl[2], currentTM.z, addAlias)));
      hw = fminf(history.w * (1.0f / 15.0f), 1.0f - curHistoryBlur);
    }
  } else {
  float gamma = current.w <= 0.0f ? 1.0f : clampf(8.0f * rcpFast(historyBlur), 1.0f, 32.0f);
  // NeighborMinMax :166-236
  float fl[4] = {currentTM.x, currentTM.y, currentTM.z, current.w};
  float nmin[4], nmax[4];
  {
    float mu[3] = {currentTM.x, currentTM.y, currentTM.z};
    const float alpha = current.w;
    float m2[3] = {mu[0] * mu[0], mu[1] * mu[1], mu[2] * mu[2]};
    for (int i = 0; i < 8; ++i) {
      const float4 t = tile[ly + oy[i]][lx + ox[i]];
      const float nb[4] = {t.x, t.y, t.z, t.w};
      const float wgt = i < 4 ? 0.5f : 0.25f;
      for (int k = 0; k < 4; ++k) fl[k] = __builtin_fmaf(nb[k], wgt, fl[k]);     // wgt is a power of two: the product is exact, fused or not
      for (int k = 0; k < 3; ++k) { mu[k] += nb[k]; m2[k] += nb[k] * nb[k]; }
    }
    for (int k = 0; k < 4; ++k) fl[k] /= 4.0f;
    gamma = fabsf(alpha - fl[3]) < 1.0f / 255.0f ? gamma : 1.0f;
    float sigma[3];
    for (int k = 0; k < 3; ++k) {
      mu[k] = divBy9(mu[k]);
      sigma[k] = sqrtRN(fabsf(divBy9(m2[k]) - mu[k] * mu[k]));
      const float gs = gamma * sigma[k];
      nmin[k] = fminf(mu[k] - gs, fl[k]);
      nmax[k] = fmaxf(mu[k] + gs, fl[k]);
    }
    nmin[3] = mu[0] - sigma[0]; nmax[3] = mu[0] + sigma[0];
  }
  // from here on smooth terms again (the clamp window nmin/nmax above is the part that needs the reference's rounding)
  {
#pragma clang fp contract(fast)
  curHistoryBlur = saturatef(curHistoryBlur);   // :290-291
  historyBlur = saturatef(historyBlur);
  const f3 hTM = tssTMSmooth(mk3(history.x, history.y, history.z));   // :294-299
  float historyTM[3] = {fminf(fmaxf(hTM.x, nmin[0]), nmax[0]), fminf(fmaxf(hTM.y, nmin[1]), nmax[1]), fminf(fmaxf(hTM.z, nmin[2]), nmax[2])};
  const float contrast = nmax[3] - nmin[3];
  const float lumContrastFactor = 32.0f * 4.0f;   // :303-308
  float addAlias = historyBlur * 0.5f + 0.25f;
  addAlias = saturatef(addAlias + rcpFast(1.0f + contrast * lumContrastFactor));
  const float ctm[3] = {currentTM.x, currentTM.y, currentTM.z};
  for (int k = 0; k < 3; ++k) fl[k] = lerpf(fl[k], ctm[k], addAlias);   // :311
  const float lumHist = historyTM[0];   // :314-325
  const float distToClamp = fminf(fabsf(nmin[3] - lumHist), fabsf(nmax[3] - lumHist));
  const float historyAmt = fminf(rcpFast(history.w) + historyBlur * 0.125f, 1.0f);
  float blend = 0.25f * rcpFast(lerpf(8.0f, distToClamp + contrast, historyAmt));
  blend = fminf(blend, 0.25f);
  blend = fl[3] > 0.0f ? blend : 1.0f;
  result = tssITM(mk3(lerpf(historyTM[0], fl[0], blend), lerpf(historyTM[1], fl[1], blend), lerpf(historyTM[2], fl[2], blend)));   // :327-329
  if (isnan(result.x) || isnan(result.y) || isnan(result.z)) result = tssITM(mk3(fl[0], fl[1], fl[2]));
  hw = fminf(history.w * (1.0f / 15.0f), 1.0f - curHistoryBlur);
  }
  }
  return packRGBA16F(result.x, result.y, result.z, hw);   // :335 (TSS[parity])
}
// Stages the tiles of a (ROWS - 2) x (PITCH - 2) pixel block whose first pixel is (ox + 1, oy + 1).
template <int ROWS, int PITCH, int THREADS>
RT_DEV void stageTemporalTiles(const Targets& T, float4 (&tile)[ROWS][PITCH], uint32_t (&velRaw)[ROWS][PITCH], float (&velSq)[ROWS][PITCH], int ox, int oy) {
  const int W = T.W, H = T.H;
  for (int t = threadIdx.x; t < ROWS * PITCH; t += THREADS) {
    const int tx = ox + t % PITCH, ty = oy + t / PITCH;
    const bool inside = tx >= 0 && ty >= 0 && tx < W && ty < H;
    const size_t ti = inside ? (size_t)ty * W + tx : 0;
    const f4 raw = inside ? unpackRGBA16F(T.fltDff[ti]) : f4{0.0f, 0.0f, 0.0f, 0.0f};
    const uint32_t vr = inside ? T.velocity[ti] : 0u;
    const f3 tm = tssTM(mk3(raw.x, raw.y, raw.z));
    tile[t / PITCH][t % PITCH] = make_float4(tm.x, tm.y, tm.z, raw.w);
    const float vx = f16ToF32(vr & 0xFFFFu), vy = f16ToF32(vr >> 16);
    velRaw[t / PITCH][t % PITCH] = vr;
    velSq[t / PITCH][t % PITCH] = vx * vx + vy * vy;
  }
}
#define RT_TP_ROWS 4
__global__ void __launch_bounds__(256) temporalKernel(Targets T) {
  __shared__ float4 tile[RT_TP_ROWS + 2][66];
  __shared__ uint32_t velRaw[RT_TP_ROWS + 2][66];      // the velocity texels of the same window (zero outside the frame) ...
  __shared__ float velSq[RT_TP_ROWS + 2][66];          // ... and their squared lengths: VelocityMax compares five of them per pixel
  stageTemporalTiles<RT_TP_ROWS + 2, 66, 256>(T, tile, velRaw, velSq, (int)blockIdx.x * 64 - 1, T.rowBegin + (int)blockIdx.y * RT_TP_ROWS - 1);
  __syncthreads();
  const int lx = (threadIdx.x & 63) + 1, ly = (threadIdx.x >> 6) + 1;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = T.rowBegin + blockIdx.y * RT_TP_ROWS + (threadIdx.x >> 6);
  if (x >= T.W || y >= T.rowEnd) return;
  T.scratch[(size_t)y * T.W + x] = temporalPixel<RT_TP_ROWS + 2, 66>(T, tile, velRaw, velSq, x, y, lx, ly);
}

// The temporal pass and the tone map in ONE kernel (round 4; CSTemporalSS.hlsl:254-336 + PSToneMap.hlsl:13-41; Denoiser.cpp:66-103 issues
// them back to back).  The tone map reads TemporalSSOut at the pixel and its four neighbours -- what the temporal pass has just written:
// as two kernels that is a launch, an event-carrying gap on the frame's longest stream and 8 bytes per pixel read again.  One workgroup
// of 1024 threads computes the temporal result of 64 x 16 pixels, one per thread (a thread that walks several surface pixels serialises
// their history taps: the 8- and 16-row shapes of round 3 lost), keeps c / (c + 0.5) of the ROUNDED result (the half-float texel the
// two-kernel path reads back: the back buffer is bit-identical) in LDS and tone-maps the 62 x 14 pixels inside: 1.18 x the temporal work.
// TemporalSSOut is written by the workgroup whose interior holds the pixel; of a strip's apron rows (b - 1 and e: what the two-kernel
// path's temporal pass writes there) by the first and last row of workgroups.
#ifndef RT_TT_TW
#define RT_TT_TW 64      // temporal pixels (= threads) of a workgroup: RT_TT_TW x RT_TT_TH; back-buffer pixels: two less each way
#define RT_TT_TH 16
#endif
#define RT_TT_W (RT_TT_TW - 2)
#define RT_TT_H (RT_TT_TH - 2)
__global__ void __launch_bounds__(RT_TT_TW * RT_TT_TH) temporalToneKernel(Targets T) {
  constexpr int TW = RT_TT_TW, TH = RT_TT_TH;
  __shared__ float4 tile[TH + 2][TW + 2];
  __shared__ uint32_t velRaw[TH + 2][TW + 2];
  __shared__ float velSq[TH + 2][TW + 2];
  __shared__ float4 tm[TH][TW];      // c / (c + 0.5) and alpha of the temporal result; zero outside the frame (what D3D reads there)
  const int x0 = (int)blockIdx.x * RT_TT_W - 1, y0 = T.outBegin + (int)blockIdx.y * RT_TT_H - 1;      // the pixel of thread (0, 0)
  stageTemporalTiles<TH + 2, TW + 2, TW * TH>(T, tile, velRaw, velSq, x0 - 1, y0 - 1);
  __syncthreads();
  const int tx = threadIdx.x % TW, ty = threadIdx.x / TW;
  const int x = x0 + tx, y = y0 + ty;
  float4 c = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (x >= 0 && x < T.W && y >= T.rowBegin && y < T.rowEnd) {      // (T.rowBegin .. T.rowEnd: the strip + one row either side, inside the frame)
    const uint2 packed = temporalPixel<TH + 2, TW + 2>(T, tile, velRaw, velSq, x, y, tx + 1, ty + 1);
    const bool mine = tx >= 1 && tx <= RT_TT_W && (ty >= 1 && ty <= RT_TT_H ? true : ty == 0 ? blockIdx.y == 0 : blockIdx.y == gridDim.y - 1);
    if (mine) T.scratch[(size_t)y * T.W + x] = packed;
    {
#pragma clang fp contract(fast)
      const f4 v = unpackRGBA16F(packed);
      c = make_float4(v.x * rcpFast(v.x + 0.5f), v.y * rcpFast(v.y + 0.5f), v.z * rcpFast(v.z + 0.5f), v.w);
    }
  }
  tm[ty][tx] = c;
  __syncthreads();
  if (tx < 1 || tx > RT_TT_W || ty < 1 || ty > RT_TT_H || x >= T.W || y >= T.outEnd) return;
  {
#pragma clang fp contract(fast)
    const float4 c0 = tm[ty][tx], c1 = tm[ty][tx - 1], c2 = tm[ty][tx + 1], c3 = tm[ty - 1][tx], c4 = tm[ty + 1][tx];
    float lx_ = -4.0f * c0.x, ly_ = -4.0f * c0.y, lz_ = -4.0f * c0.z;
    lx_ += c1.x; ly_ += c1.y; lz_ += c1.z;
    lx_ += c2.x; ly_ += c2.y; lz_ += c2.z;
    lx_ += c3.x; ly_ += c3.y; lz_ += c3.z;
    lx_ += c4.x; ly_ += c4.y; lz_ += c4.z;
    T.backbuffer[(size_t)y * T.W + x] = packRGBA8(c0.x - 0.2f * lx_, c0.y - 0.2f * ly_, c0.z - 0.2f * lz_, c0.w);
  }
}

// PSToneMap.hlsl:13-41; source = TSS[parity] (passed as T.scratch)
// One workgroup = 64 x RT_TM_ROWS pixels, RT_TM_ROWS / 4 per thread: the pass is one memory round trip and a handful of instructions per
// pixel, so a wave's life is its launch plus that round trip -- four pixels per thread put four times the bytes in flight per wave
// and take a quarter of the waves (64 x 4 blocks: 32 400 waves, 37 M wave quad-cycles per frame; profiles/r02_d_limiter.txt), and
// the one-texel apron costs 1.16 x instead of 1.55 x.
#ifndef RT_TM_ROWS
#define RT_TM_ROWS 16
#endif
__global__ void __launch_bounds__(256) toneMapKernel(Targets T) {
#pragma clang fp contract(fast)
  __shared__ float4 tile[RT_TM_ROWS + 2][66];          // c / (c + 0.5) of the block's pixels and a one-texel apron, computed once per texel
  const int W = T.W, H = T.H;
  {
    const int ox = blockIdx.x * 64 - 1, oy = T.rowBegin + blockIdx.y * RT_TM_ROWS - 1;
    constexpr int N = (RT_TM_ROWS + 2) * 66, PER = (N + 255) / 256;
    f4 c[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) { const int t = threadIdx.x + 256 * k; c[k] = t < N ? loadRGBA16(T.scratch, ox + t % 66, oy + t / 66, W, H) : f4{0.0f, 0.0f, 0.0f, 0.0f}; }      // all loads first
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int t = threadIdx.x + 256 * k;
      if (t < N) tile[t / 66][t % 66] = make_float4(c[k].x * rcpFast(c[k].x + 0.5f), c[k].y * rcpFast(c[k].y + 0.5f), c[k].z * rcpFast(c[k].z + 0.5f), c[k].w);
    }
  }
  __syncthreads();
  const int lx = (threadIdx.x & 63) + 1;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  if (x >= T.W) return;
#pragma unroll
  for (int k = 0; k < RT_TM_ROWS / 4; ++k) {
    const int ly = (threadIdx.x >> 6) + 4 * k + 1;
    const int y = T.rowBegin + blockIdx.y * RT_TM_ROWS + (threadIdx.x >> 6) + 4 * k;
    if (y >= T.rowEnd) continue;
    const float4 c0 = tile[ly][lx], c1 = tile[ly][lx - 1], c2 = tile[ly][lx + 1], c3 = tile[ly - 1][lx], c4 = tile[ly + 1][lx];
    float lx_ = -4.0f * c0.x, ly_ = -4.0f * c0.y, lz_ = -4.0f * c0.z;
    lx_ += c1.x; ly_ += c1.y; lz_ += c1.z;
    lx_ += c2.x; ly_ += c2.y; lz_ += c2.z;
    lx_ += c3.x; ly_ += c3.y; lz_ += c3.z;
    lx_ += c4.x; ly_ += c4.y; lz_ += c4.z;
    T.backbuffer[(size_t)y * W + x] = packRGBA8(c0.x - 0.2f * lx_, c0.y - 0.2f * ly_, c0.z - 0.2f * lz_, c0.w);
  }
}

static Targets makeTargets(rtggx_context* c, const FrameParams& fp, RowPass pass) {
  Targets T;
  T.normal = c->normal; T.roughMetal = c->roughMetal; T.depth32 = c->depth32; T.velocity = c->velocity;
  T.rtRefl = c->rtRefl; T.rtDiff = c->rtDiff;
  T.scratch = c->tss[c->frameParity]; T.history = c->tss[c->frameParity ^ 1u]; T.fltRfl = c->fltRfl; T.fltDff = c->fltDff; T.backbuffer = c->backbuffer;
  T.W = (int)fp.W; T.H = (int)fp.H;
  uint32_t rb, re; passRows(fp, pass, rb, re);
  T.rowBegin = (int)rb; T.rowEnd = (int)re;
  passRows(fp, ROWS_FINAL, rb, re);
  T.outBegin = (int)rb; T.outEnd = (int)re;
  const bool strip = fp.rowBegin > 0u || fp.rowEnd < fp.H;
  T.histLo = (int)(fp.rowBegin > c->historyApron ? fp.rowBegin - c->historyApron : 0u);
  T.histHi = (int)(fp.rowEnd + c->historyApron < fp.H ? fp.rowEnd + c->historyApron : fp.H);
  T.histReach = strip ? c->histReach : nullptr;
  // the other ranks' history images (rtggx_set_history_peers): the table lives in device memory, [parity][rank]
  const bool peers = strip && c->peerWorld > 1u && c->dPeerTable != nullptr;
  T.peerHist = peers ? reinterpret_cast<const uint2* const*>(c->dPeerTable) + (c->frameParity ^ 1u) * RT_MAX_PEERS : nullptr;
  T.peerBounds = peers ? reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint2* const*>(c->dPeerTable) + 2 * RT_MAX_PEERS) : nullptr;
  T.peerWorld = peers ? (int)c->peerWorld : 0;
  { uint32_t gb, ge; passRows(fp, ROWS_GBUFFER, gb, ge);      // the tiles are ray generation's
    T.tileWords = c->tileWords(gb, ge); T.tilesX = (int)((fp.W + 15) / 16); T.tileRow0 = (int)gb; }
  return T;
}

int launchDenoise(rtggx_context* c, const FrameParams& fp, int useLds, hipStream_t s, hipEvent_t done, bool fuseToneMap) {
  c->frameParity ^= 1u;   // Denoiser.cpp:69
  if (fp.rowEnd <= fp.rowBegin) return 0;
  const Targets TH = makeTargets(c, fp, ROWS_GBUFFER), TV = makeTargets(c, fp, ROWS_VFILTER), TT = makeTargets(c, fp, ROWS_TEMPORAL);
  const dim3 block(256);
  auto grid = [&](const Targets& T, uint32_t bw, uint32_t bh) { return dim3((fp.W + bw - 1) / bw, (T.rowEnd - T.rowBegin + bh - 1) / bh); };
  auto mark = [&](int i) { if (c->timing) hipEventRecord(c->tev[i], s); };
  // The diffuse passes only touch pixels whose metallic is below 1, and metallic is a per-instance material constant
  // (Material.hlsli:20-30: no textures): with both instances fully metallic -- the sample's default -- they have nothing
  // to do (the reflection V pass already wrote FilteredOut1) and are not launched.
  const bool anyDiffuse = fp.mat.RoughMetals[0][1] < 1.0f || fp.mat.RoughMetals[1][1] < 1.0f;
  // FilteredOut is read by the diffuse V pass only.  Without diffuse passes the reflection V pass writes FilteredOut1 alone (the two
  // images would be identical, 8 bytes per pixel each); rtggx_readback(RTGGX_BUF_FLT_RFL) then returns FilteredOut1 (capi.hip).
  c->fltRflIsFltDff = !anyDiffuse;
  Targets TVr = TV; if (!anyDiffuse) TVr.fltRfl = nullptr;
  if (useLds) {
    hipLaunchKernelGGL(spatialTiledKernel<0>, grid(TH, 64, 4), block, 0, s, TH); mark(4);
    hipLaunchKernelGGL(spatialTiledKernel<1>, grid(TV, RT_VBW, RT_VBH), block, 0, s, TVr); mark(5);
    if (anyDiffuse) hipLaunchKernelGGL(spatialTiledKernel<2>, grid(TH, 64, 4), block, 0, s, TH);
    mark(6);
    if (anyDiffuse) hipLaunchKernelGGL(spatialTiledKernel<3>, grid(TV, RT_VBW, RT_VBH), block, 0, s, TV);
    mark(7);
  } else if (!anyDiffuse) {
    hipLaunchKernelGGL(spatialDirectKernel<0>, grid(TH, 64, 4), block, 0, s, TH); mark(4);
    hipLaunchKernelGGL(spatialDirectKernel<1>, grid(TV, 64, 4), block, 0, s, TVr); mark(5); mark(6); mark(7);
  } else {
    hipLaunchKernelGGL(spatialDirectKernel<0>, grid(TH, 64, 4), block, 0, s, TH); mark(4);
    hipLaunchKernelGGL(spatialDirectKernel<1>, grid(TV, 64, 4), block, 0, s, TV); mark(5);
    hipLaunchKernelGGL(spatialDirectKernel<2>, grid(TH, 64, 4), block, 0, s, TH); mark(6);
    hipLaunchKernelGGL(spatialDirectKernel<3>, grid(TV, 64, 4), block, 0, s, TV); mark(7);
  }
  if (fuseToneMap) {      // the temporal pass and the tone map of its result in one kernel: workgroups of 62 x 14 back-buffer pixels over the strip's own rows
    const dim3 g((fp.W + RT_TT_W - 1) / RT_TT_W, (uint32_t)(TT.outEnd - TT.outBegin + RT_TT_H - 1) / RT_TT_H), b(RT_TT_TW * RT_TT_TH);
    if (done && c->attachEvents) hipExtLaunchKernelGGL(temporalToneKernel, g, b, 0, s, nullptr, done, 0, TT);
    else { hipLaunchKernelGGL(temporalToneKernel, g, b, 0, s, TT); if (done) hipEventRecord(done, s); }
  } else if (done && c->attachEvents) hipExtLaunchKernelGGL(temporalKernel, grid(TT, 64, RT_TP_ROWS), block, 0, s, nullptr, done, 0, TT);
  else { hipLaunchKernelGGL(temporalKernel, grid(TT, 64, RT_TP_ROWS), block, 0, s, TT); if (done) hipEventRecord(done, s); }
  mark(8);
  RT_HIP(hipGetLastError());
  return 0;
}

// The tone map as a kernel of its own: a caller that tone-maps without having denoised in this frame, the per-pass timing mode, and
// rtggx_debug_fuse_tone_map(ctx, 0) -- the two-kernel path of rounds 1-3, kept for comparison with the fused one.
int launchToneMap(rtggx_context* c, const FrameParams& fp, hipStream_t s, hipEvent_t done) {
  if (fp.rowEnd <= fp.rowBegin) return 0;
  const Targets T = makeTargets(c, fp, ROWS_FINAL);
  const dim3 grid((fp.W + 63) / 64, (uint32_t)(T.rowEnd - T.rowBegin + RT_TM_ROWS - 1) / RT_TM_ROWS), block(256);
  if (done && c->attachEvents) hipExtLaunchKernelGGL(toneMapKernel, grid, block, 0, s, nullptr, done, 0, T);
  else { hipLaunchKernelGGL(toneMapKernel, grid, block, 0, s, T); if (done) RT_HIP(hipEventRecord(done, s)); }
  RT_HIP(hipGetLastError());
  return 0;
}

}  // namespace rt
