// Environment map upload: the gfx950 side of DDS::Loader::CreateTextureFromFile
// (RayTracedGGX/Content/RayTracer.cpp:143-150) and of SphericalHarmonics::Transform
// (RayTracer.cpp:307-310, 345-350; XUSG/Advanced/XUSGAdvanced.h:623-648).
//
// D3D samples BC6H in texture hardware; CDNA4 has no BC6H unit, so the blocks are decoded once
// into an RGBA16F mip pyramid (mip-major, six faces per mip) that the trace kernel filters by
// hand.  BC6H_UF16 / BC6H_SF16 decode follows the published format (Direct3D 11 "BC6H format" / Khronos Data
// Format Specification, BPTC float): one lane per 4x4 block.
// SH projection: one lane per mip-0 texel, solid-angle weighted, real orthonormal basis in the
// consumer's axis convention (SHIrradianceTypeless.hlsli:23-25: x=-n.x, y=-n.y, z=n.z),
// workgroup reduction in LDS, fp64 atomics; weights normalised to 4*pi.
#include <vector>
#include <cstring>
#include "rtggx_context.h"

namespace rt {

// ---- BC6H mode descriptors -----------------------------------------------------------------------------
// Fields: 0 m, 1 d, then r/g/b x w/x/y/z = 2 + 4*channel + endpoint.
struct Bc6Mode { uint8_t id, transformed, regions, wbits, delta[3]; uint8_t nbits; uint8_t field[82], bit[82]; };

// Header bit streams, least significant bit first.  "f[a:b]" with a>=b is bits b..a ascending,
// with a<b it is bits b..a descending (the reversed high bits of modes 13 and 14).
static const struct { uint8_t id, transformed, regions, wbits, dr, dg, db; const char* bits; } kModeSrc[14] = {
  {0x00, 1, 2, 10, 5, 5, 5, "m1:0 gy4 by4 bz4 rw9:0 gw9:0 bw9:0 rx4:0 gz4 gy3:0 gx4:0 bz0 gz3:0 bx4:0 bz1 by3:0 ry4:0 bz2 rz4:0 bz3 d4:0"},
  {0x01, 1, 2, 7, 6, 6, 6, "m1:0 gy5 gz4 gz5 rw6:0 bz0 bz1 by4 gw6:0 by5 bz2 gy4 bw6:0 bz3 bz5 bz4 rx5:0 gy3:0 gx5:0 gz3:0 bx5:0 by3:0 ry5:0 rz5:0 d4:0"},
  {0x02, 1, 2, 11, 5, 4, 4, "m4:0 rw9:0 gw9:0 bw9:0 rx4:0 rw10 gy3:0 gx3:0 gw10 bz0 gz3:0 bx3:0 bw10 bz1 by3:0 ry4:0 bz2 rz4:0 bz3 d4:0"},
  {0x06, 1, 2, 11, 4, 5, 4, "m4:0 rw9:0 gw9:0 bw9:0 rx3:0 rw10 gz4 gy3:0 gx4:0 gw10 gz3:0 bx3:0 bw10 bz1 by3:0 ry3:0 bz0 bz2 rz3:0 gy4 bz3 d4:0"},
  {0x0a, 1, 2, 11, 4, 4, 5, "m4:0 rw9:0 gw9:0 bw9:0 rx3:0 rw10 by4 gy3:0 gx3:0 gw10 bz0 gz3:0 bx4:0 bw10 by3:0 ry3:0 bz1 bz2 rz3:0 bz4 bz3 d4:0"},
  {0x0e, 1, 2, 9, 5, 5, 5, "m4:0 rw8:0 by4 gw8:0 gy4 bw8:0 bz4 rx4:0 gz4 gy3:0 gx4:0 bz0 gz3:0 bx4:0 bz1 by3:0 ry4:0 bz2 rz4:0 bz3 d4:0"},
  {0x12, 1, 2, 8, 6, 5, 5, "m4:0 rw7:0 gz4 by4 gw7:0 bz2 gy4 bw7:0 bz3 bz4 rx5:0 gy3:0 gx4:0 bz0 gz3:0 bx4:0 bz1 by3:0 ry5:0 rz5:0 d4:0"},
  {0x16, 1, 2, 8, 5, 6, 5, "m4:0 rw7:0 bz0 by4 gw7:0 gy5 gy4 bw7:0 gz5 bz4 rx4:0 gz4 gy3:0 gx5:0 gz3:0 bx4:0 bz1 by3:0 ry4:0 bz2 rz4:0 bz3 d4:0"},
  {0x1a, 1, 2, 8, 5, 5, 6, "m4:0 rw7:0 bz1 by4 gw7:0 by5 gy4 bw7:0 bz5 bz4 rx4:0 gz4 gy3:0 gx4:0 bz0 gz3:0 bx5:0 by3:0 ry4:0 bz2 rz4:0 bz3 d4:0"},
  {0x1e, 0, 2, 6, 6, 6, 6, "m4:0 rw5:0 gz4 bz0 bz1 by4 gw5:0 gy5 by5 bz2 gy4 bw5:0 gz5 bz3 bz5 bz4 rx5:0 gy3:0 gx5:0 gz3:0 bx5:0 by3:0 ry5:0 rz5:0 d4:0"},
  {0x03, 0, 1, 10, 10, 10, 10, "m4:0 rw9:0 gw9:0 bw9:0 rx9:0 gx9:0 bx9:0"},
  {0x07, 1, 1, 11, 9, 9, 9, "m4:0 rw9:0 gw9:0 bw9:0 rx8:0 rw10 gx8:0 gw10 bx8:0 bw10"},
  {0x0b, 1, 1, 12, 8, 8, 8, "m4:0 rw9:0 gw9:0 bw9:0 rx7:0 rw10:11 gx7:0 gw10:11 bx7:0 bw10:11"},
  {0x0f, 1, 1, 16, 4, 4, 4, "m4:0 rw9:0 gw9:0 bw9:0 rx3:0 rw10:15 gx3:0 gw10:15 bx3:0 bw10:15"},
};

static void buildModeTable(std::vector<Bc6Mode>& out) {
  out.resize(14);
  for (int mi = 0; mi < 14; ++mi) {
    Bc6Mode& m = out[mi];
    std::memset(&m, 0, sizeof m);
    m.id = kModeSrc[mi].id; m.transformed = kModeSrc[mi].transformed; m.regions = kModeSrc[mi].regions; m.wbits = kModeSrc[mi].wbits;
    m.delta[0] = kModeSrc[mi].dr; m.delta[1] = kModeSrc[mi].dg; m.delta[2] = kModeSrc[mi].db;
    const char* s = kModeSrc[mi].bits;
    int n = 0;
    while (*s) {
      while (*s == ' ') ++s;
      if (!*s) break;
      int field;
      if (*s == 'm') { field = 0; ++s; } else if (*s == 'd') { field = 1; ++s; }
      else { const int ch = *s == 'r' ? 0 : (*s == 'g' ? 1 : 2); const int ep = s[1] - 'w'; field = 2 + 4 * ch + ep; s += 2; }
      int a = 0; while (*s >= '0' && *s <= '9') a = a * 10 + (*s++ - '0');
      int b = a;
      if (*s == ':') { ++s; b = 0; while (*s >= '0' && *s <= '9') b = b * 10 + (*s++ - '0'); }
      if (a >= b) for (int k = b; k <= a; ++k) { m.field[n] = (uint8_t)field; m.bit[n] = (uint8_t)k; ++n; }
      else for (int k = b; k >= a; --k) { m.field[n] = (uint8_t)field; m.bit[n] = (uint8_t)k; ++n; }
    }
    m.nbits = (uint8_t)n;
  }
}

__constant__ uint8_t kPartition2[32][16] = {
  {0,0,1,1,0,0,1,1,0,0,1,1,0,0,1,1}, {0,0,0,1,0,0,0,1,0,0,0,1,0,0,0,1}, {0,1,1,1,0,1,1,1,0,1,1,1,0,1,1,1}, {0,0,0,1,0,0,1,1,0,0,1,1,0,1,1,1},
  {0,0,0,0,0,0,0,1,0,0,0,1,0,0,1,1}, {0,0,1,1,0,1,1,1,0,1,1,1,1,1,1,1}, {0,0,0,1,0,0,1,1,0,1,1,1,1,1,1,1}, {0,0,0,0,0,0,0,1,0,0,1,1,0,1,1,1},
  {0,0,0,0,0,0,0,0,0,0,0,1,0,0,1,1}, {0,0,1,1,0,1,1,1,1,1,1,1,1,1,1,1}, {0,0,0,0,0,0,0,1,0,1,1,1,1,1,1,1}, {0,0,0,0,0,0,0,0,0,0,0,1,0,1,1,1},
  {0,0,0,1,0,1,1,1,1,1,1,1,1,1,1,1}, {0,0,0,0,0,0,0,0,1,1,1,1,1,1,1,1}, {0,0,0,0,1,1,1,1,1,1,1,1,1,1,1,1}, {0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1},
  {0,0,0,0,1,0,0,0,1,1,1,0,1,1,1,1}, {0,1,1,1,0,0,0,1,0,0,0,0,0,0,0,0}, {0,0,0,0,0,0,0,0,1,0,0,0,1,1,1,0}, {0,1,1,1,0,0,1,1,0,0,0,1,0,0,0,0},
  {0,0,1,1,0,0,0,1,0,0,0,0,0,0,0,0}, {0,0,0,0,1,0,0,0,1,1,0,0,1,1,1,0}, {0,0,0,0,0,0,0,0,1,0,0,0,1,1,0,0}, {0,1,1,1,0,0,1,1,0,0,1,1,0,0,0,1},
  {0,0,1,1,0,0,0,1,0,0,0,1,0,0,0,0}, {0,0,0,0,1,0,0,0,1,0,0,0,1,1,0,0}, {0,1,1,0,0,1,1,0,0,1,1,0,0,1,1,0}, {0,0,1,1,0,1,1,0,0,1,1,0,1,1,0,0},
  {0,0,0,1,0,1,1,1,1,1,1,0,1,0,0,0}, {0,0,0,0,1,1,1,1,1,1,1,1,0,0,0,0}, {0,1,1,1,0,0,0,1,1,0,0,0,1,1,1,0}, {0,0,1,1,1,0,0,1,1,0,0,1,1,1,0,0}};
__constant__ uint8_t kAnchor2[32] = {15,15,15,15,15,15,15,15, 15,15,15,15,15,15,15,15, 15,2,8,2,2,8,8,15, 2,8,2,2,8,8,2,2};
__constant__ int kWeight3[8] = {0, 9, 18, 27, 37, 46, 55, 64};
__constant__ int kWeight4[16] = {0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64};

RT_DEV int blockBit(const uint32_t blk[4], int pos) { return (blk[pos >> 5] >> (pos & 31)) & 1; }
RT_DEV int unquantizeU(int c, int bits) {
  if (bits >= 15) return c;
  if (c == 0) return 0;
  if (c == (1 << bits) - 1) return 0xFFFF;
  return ((c << 16) + 0x8000) >> bits;
}

// BC6H_SF16: endpoints are two's-complement numbers of wbits bits; magnitudes unquantise to 15 bits, the sign is carried
RT_DEV int signExtendBits(int v, int bits) { const int m = 1 << (bits - 1); return ((v & ((1 << bits) - 1)) ^ m) - m; }
RT_DEV int unquantizeS(int c, int bits) {
  if (bits >= 16) return c;
  const bool neg = c < 0;
  if (neg) c = -c;
  int u;
  if (c == 0) u = 0;
  else if (c >= (1 << (bits - 1)) - 1) u = 0x7FFF;
  else u = ((c << 15) + 0x4000) >> (bits - 1);
  return neg ? -u : u;
}

// One lane decodes one block of one mip of one face and writes up to 16 RGBA16F texels.
__global__ void bc6hDecodeKernel(const uint32_t* __restrict__ blocks, const Bc6Mode* __restrict__ modes, uint2* __restrict__ dst,
                                 uint32_t size, uint32_t blocksPerRow, uint32_t numBlocks, int isSigned) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= numBlocks) return;
  uint32_t blk[4];
  for (int k = 0; k < 4; ++k) blk[k] = blocks[4 * (size_t)b + k];
  int modeBits = blk[0] & 3;
  if (modeBits >= 2) modeBits = blk[0] & 31;
  int mi = -1;
  for (int k = 0; k < 14; ++k) if (modes[k].id == modeBits) mi = k;
  const uint32_t bx = b % blocksPerRow, by = b / blocksPerRow;
  uint16_t px[16][3];
  if (mi < 0) { for (int i = 0; i < 16; ++i) px[i][0] = px[i][1] = px[i][2] = 0; }
  else {
    const Bc6Mode& md = modes[mi];
    int fld[14];
    for (int k = 0; k < 14; ++k) fld[k] = 0;
    for (int i = 0; i < md.nbits; ++i) fld[md.field[i]] |= blockBit(blk, i) << md.bit[i];
    int e[2][2][3];
    for (int c = 0; c < 3; ++c) {
      const int w = fld[2 + 4 * c], x = fld[3 + 4 * c], y = fld[4 + 4 * c], z = fld[5 + 4 * c];
      e[0][0][c] = w;
      if (md.transformed) {
        const int mask = (1 << md.wbits) - 1, sb = 1 << (md.delta[c] - 1);
        e[0][1][c] = (w + ((x ^ sb) - sb)) & mask;
        e[1][0][c] = (w + ((y ^ sb) - sb)) & mask;
        e[1][1][c] = (w + ((z ^ sb) - sb)) & mask;
      } else { e[0][1][c] = x; e[1][0][c] = y; e[1][1][c] = z; }
      if (isSigned) for (int r = 0; r < 2; ++r) for (int k = 0; k < 2; ++k) e[r][k][c] = signExtendBits(e[r][k][c], md.wbits);
    }
    for (int r = 0; r < 2; ++r) for (int k = 0; k < 2; ++k) for (int c = 0; c < 3; ++c)
      e[r][k][c] = isSigned ? unquantizeS(e[r][k][c], md.wbits) : unquantizeU(e[r][k][c], md.wbits);
    const int part = md.regions == 2 ? fld[1] : 0;
    const int ibits = md.regions == 2 ? 3 : 4;
    int pos = md.regions == 2 ? 82 : 65;
    for (int i = 0; i < 16; ++i) {
      const int region = md.regions == 2 ? kPartition2[part][i] : 0;
      const bool anchor = (i == 0) || (md.regions == 2 && i == kAnchor2[part]);
      const int n = anchor ? ibits - 1 : ibits;
      int idx = 0;
      for (int k = 0; k < n; ++k) idx |= blockBit(blk, pos + k) << k;
      pos += n;
      const int wgt = ibits == 3 ? kWeight3[idx] : kWeight4[idx];
      for (int c = 0; c < 3; ++c) {
        const int v = (e[region][0][c] * (64 - wgt) + e[region][1][c] * wgt + 32) >> 6;
        if (!isSigned) px[i][c] = (uint16_t)((v * 31) >> 6);
        else { const int m = v < 0 ? ((-v) * 31) >> 5 : (v * 31) >> 5; px[i][c] = (uint16_t)(v < 0 ? (0x8000 | m) : m); }
      }
    }
  }
  for (uint32_t y = 0; y < 4; ++y) for (uint32_t x = 0; x < 4; ++x) {
    const uint32_t X = bx * 4 + x, Y = by * 4 + y;
    if (X >= size || Y >= size) continue;
    const uint16_t* p = px[y * 4 + x];
    dst[(size_t)Y * size + X] = make_uint2((uint32_t)p[0] | ((uint32_t)p[1] << 16), (uint32_t)p[2] | (0x3C00u << 16));
  }
}
__global__ void f32ToF16Kernel(const float* __restrict__ src, uint2* __restrict__ dst, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = packRGBA16F(src[4 * (size_t)i], src[4 * (size_t)i + 1], src[4 * (size_t)i + 2], src[4 * (size_t)i + 3]);
}

int decodeEnv(rtggx_context* c, int format, uint32_t size, uint32_t mips, const void* hostData, size_t bytes, hipStream_t s) {
  // cubes up to 8192^2 (the reference's maxsize, RayTracer.cpp:146) with at most a full mip chain: log2(size) + 1 <= 14 levels
  if (size == 0 || size > 8192 || mips == 0 || mips > 14 || (size >> (mips - 1)) == 0) { setError("rtggx_set_env: bad size/mips (size %u, %u mips)", size, mips); return -1; }
  size_t perFace = 0; uint64_t texels = 0;
  for (uint32_t m = 0; m < mips; ++m) {
    const uint32_t sz = size >> m;
    perFace += (format == RTGGX_FORMAT_BC6H_UF16 || format == RTGGX_FORMAT_BC6H_SF16) ? (size_t)((sz + 3) / 4) * ((sz + 3) / 4) * 16 : (size_t)sz * sz * (format == RTGGX_FORMAT_RGBA16F ? 8 : 16);
    c->env.mipOffset[m] = (uint32_t)texels;
    texels += 6ull * sz * sz;
  }
  if (format != RTGGX_FORMAT_BC6H_UF16 && format != RTGGX_FORMAT_BC6H_SF16 && format != RTGGX_FORMAT_RGBA16F && format != RTGGX_FORMAT_RGBA32F) { setError("rtggx_set_env: unsupported format %d", format); return -1; }
  if (bytes < perFace * 6) { setError("rtggx_set_env: %zu bytes given, %zu needed", bytes, perFace * 6); return -1; }
  if (c->env.texels) { RT_HIP(hipFree(c->env.texels)); c->env.texels = nullptr; }
  RT_HIP(hipMalloc(&c->env.texels, texels * sizeof(uint2)));
  c->env.size = size; c->env.mips = mips; c->env.totalTexels = texels;
  void* dSrc = nullptr;
  RT_HIP(hipMalloc(&dSrc, perFace * 6));
  RT_HIP(hipMemcpyAsync(dSrc, hostData, perFace * 6, hipMemcpyHostToDevice, s));
  Bc6Mode* dModes = nullptr;
  const bool bc6h = format == RTGGX_FORMAT_BC6H_UF16 || format == RTGGX_FORMAT_BC6H_SF16;
  if (bc6h) {
    std::vector<Bc6Mode> modes; buildModeTable(modes);
    RT_HIP(hipMalloc(&dModes, sizeof(Bc6Mode) * 14));
    RT_HIP(hipMemcpyAsync(dModes, modes.data(), sizeof(Bc6Mode) * 14, hipMemcpyHostToDevice, s));
    RT_HIP(hipStreamSynchronize(s));   // `modes` is about to go out of scope
  }
  for (uint32_t face = 0; face < 6; ++face) {
    size_t off = perFace * face;
    for (uint32_t m = 0; m < mips; ++m) {
      const uint32_t sz = size >> m;
      uint2* dst = c->env.texels + c->env.mipOffset[m] + (size_t)face * sz * sz;
      if (bc6h) {
        const uint32_t bpr = (sz + 3) / 4, nb = bpr * bpr;
        hipLaunchKernelGGL(bc6hDecodeKernel, dim3((nb + 63) / 64), dim3(64), 0, s, (const uint32_t*)((const char*)dSrc + off), dModes, dst, sz, bpr, nb, format == RTGGX_FORMAT_BC6H_SF16 ? 1 : 0);
        off += (size_t)nb * 16;
      } else if (format == RTGGX_FORMAT_RGBA16F) {
        RT_HIP(hipMemcpyAsync(dst, (const char*)dSrc + off, (size_t)sz * sz * 8, hipMemcpyDeviceToDevice, s));
        off += (size_t)sz * sz * 8;
      } else {
        hipLaunchKernelGGL(f32ToF16Kernel, dim3((sz * sz + 255) / 256), dim3(256), 0, s, (const float*)((const char*)dSrc + off), dst, sz * sz);
        off += (size_t)sz * sz * 16;
      }
    }
  }
  RT_HIP(hipGetLastError());
  RT_HIP(hipStreamSynchronize(s));
  hipFree(dSrc);
  if (dModes) hipFree(dModes);
  RT_HIP(hipMemcpy(c->dEnvMipOffset, c->env.mipOffset, 16 * sizeof(uint32_t), hipMemcpyHostToDevice));
  c->sceneDirty = true; c->shDone = false;
  return 0;
}

// ---- SH projection ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) shProjectKernel(const uint2* __restrict__ texels, uint32_t size, double* __restrict__ acc) {
  __shared__ double red[256];
  const uint32_t n = 6u * size * size;
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  double v[28];
  for (int k = 0; k < 28; ++k) v[k] = 0.0;
  if (i < n) {
    const uint32_t face = i / (size * size), rem = i % (size * size), y = rem / size, x = rem % size;
    const double u = ((double)x + 0.5) / (double)size * 2.0 - 1.0, w = ((double)y + 0.5) / (double)size * 2.0 - 1.0;
    double dx, dy, dz;
    switch (face) {
      case 0: dx = 1.0; dy = -w; dz = -u; break;
      case 1: dx = -1.0; dy = -w; dz = u; break;
      case 2: dx = u; dy = 1.0; dz = w; break;
      case 3: dx = u; dy = -1.0; dz = -w; break;
      case 4: dx = u; dy = -w; dz = 1.0; break;
      default: dx = -u; dy = -w; dz = -1.0; break;
    }
    const double len = sqrt(dx * dx + dy * dy + dz * dz);
    const double sx = -dx / len, sy = -dy / len, sz = dz / len;
    const double wt = 1.0 / (len * len * len);
    const double Y[9] = {0.28209479177387814, 0.4886025119029199 * sy, 0.4886025119029199 * sz, 0.4886025119029199 * sx,
                         1.0925484305920792 * sx * sy, 1.0925484305920792 * sy * sz, 0.31539156525252005 * (3.0 * sz * sz - 1.0),
                         1.0925484305920792 * sx * sz, 0.5462742152960396 * (sx * sx - sy * sy)};
    const uint2 t = texels[i];
    const double L[3] = {(double)f16ToF32(t.x & 0xFFFFu), (double)f16ToF32(t.x >> 16), (double)f16ToF32(t.y & 0xFFFFu)};
    for (int k = 0; k < 9; ++k) for (int ch = 0; ch < 3; ++ch) v[3 * k + ch] = Y[k] * wt * L[ch];
    v[27] = wt;
  }
  for (int k = 0; k < 28; ++k) {
    red[threadIdx.x] = v[k];
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) { if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st]; __syncthreads(); }
    if (threadIdx.x == 0) atomicAdd(&acc[k], red[0]);
    __syncthreads();
  }
}
__global__ void shFinalizeKernel(const double* __restrict__ acc, float* __restrict__ sh) {
  const int k = threadIdx.x;
  if (k < 27) sh[k] = (float)(acc[k] * (4.0 * 3.14159265358979323846 / acc[27]));
}

int projectSH(rtggx_context* c, hipStream_t s) {
  if (!c->env.texels) { setError("rtggx_transform_sh: no environment map"); return -1; }
  double* acc = nullptr;
  RT_HIP(hipMalloc(&acc, 28 * sizeof(double)));
  RT_HIP(hipMemsetAsync(acc, 0, 28 * sizeof(double), s));
  const uint32_t n = 6u * c->env.size * c->env.size;
  hipLaunchKernelGGL(shProjectKernel, dim3((n + 255) / 256), dim3(256), 0, s, c->env.texels, c->env.size, acc);
  hipLaunchKernelGGL(shFinalizeKernel, dim3(1), dim3(64), 0, s, acc, c->sh);
  RT_HIP(hipGetLastError());
  RT_HIP(hipStreamSynchronize(s));
  hipFree(acc);
  c->shDone = true;
  return 0;
}

}  // namespace rt
