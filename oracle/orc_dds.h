// ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked, imported or called by the product path.
// DDS cube-map container parse + BC6H_UF16 block decode.  The reference hands the file to
// XUSG's DDS::Loader (RayTracer.cpp:143-150; body in the closed XUSG.dll) and the decode happens
// in texture hardware, so neither is in the reference tree: "parity unpinned" (SURVEY.md 8c).
// BC6H is a public bit-exact format; this is a restatement of its published decode procedure
// (Direct3D 11 "BC6H format" documentation / Khronos Data Format Specification, BPTC float):
// 14 modes, 1 or 2 regions, 32 two-region partitions, unsigned (UF16) and signed (SF16) unquantisation,
// 6-bit interpolation weights, final * 31/64 scaling to a binary16 bit pattern.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

namespace orc {

namespace bc6h {
// field ids
enum { M = 0, D, RW, RX, RY, RZ, GW, GX, GY, GZ, BW, BX, BY, BZ, NF };
struct Bit { uint8_t f, b; };
struct Mode { uint8_t id, transformed, regions, wbits, dr, dg, db; const char* layout; };

// Bit layouts, LSB first; "f[hi:lo]" lists bits lo..hi in increasing stream position,
// "f[lo:hi]" (lo<hi written first) lists them in decreasing order -- as the published tables do.
static const Mode kModes[14] = {
  {0x00, 1, 2, 10, 5, 5, 5, "m[1:0] gy[4] by[4] bz[4] rw[9:0] gw[9:0] bw[9:0] rx[4:0] gz[4] gy[3:0] gx[4:0] bz[0] gz[3:0] bx[4:0] bz[1] by[3:0] ry[4:0] bz[2] rz[4:0] bz[3] d[4:0]"},
  {0x01, 1, 2, 7, 6, 6, 6, "m[1:0] gy[5] gz[4] gz[5] rw[6:0] bz[0] bz[1] by[4] gw[6:0] by[5] bz[2] gy[4] bw[6:0] bz[3] bz[5] bz[4] rx[5:0] gy[3:0] gx[5:0] gz[3:0] bx[5:0] by[3:0] ry[5:0] rz[5:0] d[4:0]"},
  {0x02, 1, 2, 11, 5, 4, 4, "m[4:0] rw[9:0] gw[9:0] bw[9:0] rx[4:0] rw[10] gy[3:0] gx[3:0] gw[10] bz[0] gz[3:0] bx[3:0] bw[10] bz[1] by[3:0] ry[4:0] bz[2] rz[4:0] bz[3] d[4:0]"},
  {0x06, 1, 2, 11, 4, 5, 4, "m[4:0] rw[9:0] gw[9:0] bw[9:0] rx[3:0] rw[10] gz[4] gy[3:0] gx[4:0] gw[10] gz[3:0] bx[3:0] bw[10] bz[1] by[3:0] ry[3:0] bz[0] bz[2] rz[3:0] gy[4] bz[3] d[4:0]"},
  {0x0a, 1, 2, 11, 4, 4, 5, "m[4:0] rw[9:0] gw[9:0] bw[9:0] rx[3:0] rw[10] by[4] gy[3:0] gx[3:0] gw[10] bz[0] gz[3:0] bx[4:0] bw[10] by[3:0] ry[3:0] bz[1] bz[2] rz[3:0] bz[4] bz[3] d[4:0]"},
  {0x0e, 1, 2, 9, 5, 5, 5, "m[4:0] rw[8:0] by[4] gw[8:0] gy[4] bw[8:0] bz[4] rx[4:0] gz[4] gy[3:0] gx[4:0] bz[0] gz[3:0] bx[4:0] bz[1] by[3:0] ry[4:0] bz[2] rz[4:0] bz[3] d[4:0]"},
  {0x12, 1, 2, 8, 6, 5, 5, "m[4:0] rw[7:0] gz[4] by[4] gw[7:0] bz[2] gy[4] bw[7:0] bz[3] bz[4] rx[5:0] gy[3:0] gx[4:0] bz[0] gz[3:0] bx[4:0] bz[1] by[3:0] ry[5:0] rz[5:0] d[4:0]"},
  {0x16, 1, 2, 8, 5, 6, 5, "m[4:0] rw[7:0] bz[0] by[4] gw[7:0] gy[5] gy[4] bw[7:0] gz[5] bz[4] rx[4:0] gz[4] gy[3:0] gx[5:0] gz[3:0] bx[4:0] bz[1] by[3:0] ry[4:0] bz[2] rz[4:0] bz[3] d[4:0]"},
  {0x1a, 1, 2, 8, 5, 5, 6, "m[4:0] rw[7:0] bz[1] by[4] gw[7:0] by[5] gy[4] bw[7:0] bz[5] bz[4] rx[4:0] gz[4] gy[3:0] gx[4:0] bz[0] gz[3:0] bx[5:0] by[3:0] ry[4:0] bz[2] rz[4:0] bz[3] d[4:0]"},
  {0x1e, 0, 2, 6, 6, 6, 6, "m[4:0] rw[5:0] gz[4] bz[0] bz[1] by[4] gw[5:0] gy[5] by[5] bz[2] gy[4] bw[5:0] gz[5] bz[3] bz[5] bz[4] rx[5:0] gy[3:0] gx[5:0] gz[3:0] bx[5:0] by[3:0] ry[5:0] rz[5:0] d[4:0]"},
  {0x03, 0, 1, 10, 10, 10, 10, "m[4:0] rw[9:0] gw[9:0] bw[9:0] rx[9:0] gx[9:0] bx[9:0]"},
  {0x07, 1, 1, 11, 9, 9, 9, "m[4:0] rw[9:0] gw[9:0] bw[9:0] rx[8:0] rw[10] gx[8:0] gw[10] bx[8:0] bw[10]"},
  {0x0b, 1, 1, 12, 8, 8, 8, "m[4:0] rw[9:0] gw[9:0] bw[9:0] rx[7:0] rw[10:11] gx[7:0] gw[10:11] bx[7:0] bw[10:11]"},
  {0x0f, 1, 1, 16, 4, 4, 4, "m[4:0] rw[9:0] gw[9:0] bw[9:0] rx[3:0] rw[10:15] gx[3:0] gw[10:15] bx[3:0] bw[10:15]"},
};
static const uint8_t kPartition2[32][16] = {
  {0,0,1,1,0,0,1,1,0,0,1,1,0,0,1,1}, {0,0,0,1,0,0,0,1,0,0,0,1,0,0,0,1}, {0,1,1,1,0,1,1,1,0,1,1,1,0,1,1,1}, {0,0,0,1,0,0,1,1,0,0,1,1,0,1,1,1},
  {0,0,0,0,0,0,0,1,0,0,0,1,0,0,1,1}, {0,0,1,1,0,1,1,1,0,1,1,1,1,1,1,1}, {0,0,0,1,0,0,1,1,0,1,1,1,1,1,1,1}, {0,0,0,0,0,0,0,1,0,0,1,1,0,1,1,1},
  {0,0,0,0,0,0,0,0,0,0,0,1,0,0,1,1}, {0,0,1,1,0,1,1,1,1,1,1,1,1,1,1,1}, {0,0,0,0,0,0,0,1,0,1,1,1,1,1,1,1}, {0,0,0,0,0,0,0,0,0,0,0,1,0,1,1,1},
  {0,0,0,1,0,1,1,1,1,1,1,1,1,1,1,1}, {0,0,0,0,0,0,0,0,1,1,1,1,1,1,1,1}, {0,0,0,0,1,1,1,1,1,1,1,1,1,1,1,1}, {0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1},
  {0,0,0,0,1,0,0,0,1,1,1,0,1,1,1,1}, {0,1,1,1,0,0,0,1,0,0,0,0,0,0,0,0}, {0,0,0,0,0,0,0,0,1,0,0,0,1,1,1,0}, {0,1,1,1,0,0,1,1,0,0,0,1,0,0,0,0},
  {0,0,1,1,0,0,0,1,0,0,0,0,0,0,0,0}, {0,0,0,0,1,0,0,0,1,1,0,0,1,1,1,0}, {0,0,0,0,0,0,0,0,1,0,0,0,1,1,0,0}, {0,1,1,1,0,0,1,1,0,0,1,1,0,0,0,1},
  {0,0,1,1,0,0,0,1,0,0,0,1,0,0,0,0}, {0,0,0,0,1,0,0,0,1,0,0,0,1,1,0,0}, {0,1,1,0,0,1,1,0,0,1,1,0,0,1,1,0}, {0,0,1,1,0,1,1,0,0,1,1,0,1,1,0,0},
  {0,0,0,1,0,1,1,1,1,1,1,0,1,0,0,0}, {0,0,0,0,1,1,1,1,1,1,1,1,0,0,0,0}, {0,1,1,1,0,0,0,1,1,0,0,0,1,1,1,0}, {0,0,1,1,1,0,0,1,1,0,0,1,1,1,0,0}};
static const uint8_t kAnchor2[32] = {15,15,15,15,15,15,15,15, 15,15,15,15,15,15,15,15, 15,2,8,2,2,8,8,15, 2,8,2,2,8,8,2,2};
static const int kW3[8] = {0, 9, 18, 27, 37, 46, 55, 64};
static const int kW4[16] = {0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64};

static inline int fieldId(const char* s) {
  if (s[0] == 'm') return M; if (s[0] == 'd') return D;
  const int c = s[0] == 'r' ? 0 : (s[0] == 'g' ? 1 : 2);
  const int e = s[1] - 'w';
  return RW + c * 4 + e;
}
// Expand a layout string into per-stream-bit (field, bit) pairs.
static inline int parseLayout(const char* s, Bit* out) {
  int n = 0;
  while (*s) {
    while (*s == ' ') ++s;
    if (!*s) break;
    const int f = fieldId(s);
    while (*s != '[') ++s;
    ++s;
    int a = 0; while (*s >= '0' && *s <= '9') a = a * 10 + (*s++ - '0');
    int b = a;
    if (*s == ':') { ++s; b = 0; while (*s >= '0' && *s <= '9') b = b * 10 + (*s++ - '0'); }
    ++s;  // ']'
    // "hi:lo" -> stream order lo..hi ; "lo:hi" (reversed notation) -> stream order hi..lo
    if (a >= b) for (int k = b; k <= a; ++k) out[n++] = {(uint8_t)f, (uint8_t)k};
    else for (int k = b; k >= a; --k) out[n++] = {(uint8_t)f, (uint8_t)k};
  }
  return n;
}
static inline int getBit(const uint8_t* blk, int pos) { return (blk[pos >> 3] >> (pos & 7)) & 1; }
static inline int signExtend(int v, int bits) { const int m = 1 << (bits - 1); return (v ^ m) - m; }
static inline int unquantize(int c, int bits) {
  if (bits >= 15) return c;
  if (c == 0) return 0;
  if (c == (1 << bits) - 1) return 0xFFFF;
  return ((c << 16) + 0x8000) >> bits;
}

// BC6H_SF16 ("BC6H format", signed): endpoints are two's-complement numbers of wbits bits; magnitudes unquantise to 15 bits
static inline int unquantizeSigned(int c, int bits) {
  if (bits >= 16) return c;
  const bool neg = c < 0;
  if (neg) c = -c;
  int u;
  if (c == 0) u = 0;
  else if (c >= (1 << (bits - 1)) - 1) u = 0x7FFF;
  else u = ((c << 15) + 0x4000) >> (bits - 1);
  return neg ? -u : u;
}

// Decode one 16-byte block into 16 texels x 3 binary16 bit patterns (row-major 4x4).
static inline void decodeBlock(const uint8_t* blk, uint16_t out[16][3], bool isSigned = false) {
  int modeBits = blk[0] & 3;
  if (modeBits >= 2) modeBits = blk[0] & 31;
  const Mode* md = nullptr;
  for (const Mode& m : kModes) if (m.id == modeBits) md = &m;
  if (!md) { std::memset(out, 0, 16 * 3 * 2); return; }   // reserved modes decode to zero
  Bit bits[96];
  const int nb = parseLayout(md->layout, bits);
  int fld[NF] = {0};
  for (int i = 0; i < nb; ++i) fld[bits[i].f] |= getBit(blk, i) << bits[i].b;
  int e[2][2][3];   // [region][endpoint][channel]
  const int delta[3] = {md->dr, md->dg, md->db};
  for (int c = 0; c < 3; ++c) {
    const int w = fld[RW + 4 * c], x = fld[RX + 4 * c], y = fld[RY + 4 * c], z = fld[RZ + 4 * c];
    e[0][0][c] = w;
    if (md->transformed) {
      const int mask = (1 << md->wbits) - 1;
      e[0][1][c] = (w + signExtend(x, delta[c])) & mask;
      e[1][0][c] = (w + signExtend(y, delta[c])) & mask;
      e[1][1][c] = (w + signExtend(z, delta[c])) & mask;
    } else { e[0][1][c] = x; e[1][0][c] = y; e[1][1][c] = z; }
    if (isSigned) for (int r = 0; r < 2; ++r) for (int k = 0; k < 2; ++k) e[r][k][c] = signExtend(e[r][k][c] & ((1 << md->wbits) - 1), md->wbits);
  }
  for (int r = 0; r < 2; ++r) for (int k = 0; k < 2; ++k) for (int c = 0; c < 3; ++c)
    e[r][k][c] = isSigned ? unquantizeSigned(e[r][k][c], md->wbits) : unquantize(e[r][k][c], md->wbits);
  const int part = md->regions == 2 ? fld[D] : 0;
  const int ibits = md->regions == 2 ? 3 : 4;
  int pos = md->regions == 2 ? 82 : 65;
  for (int i = 0; i < 16; ++i) {
    const int region = md->regions == 2 ? kPartition2[part][i] : 0;
    const bool anchor = (i == 0) || (md->regions == 2 && i == kAnchor2[part]);
    const int n = anchor ? ibits - 1 : ibits;
    int idx = 0;
    for (int k = 0; k < n; ++k) idx |= getBit(blk, pos + k) << k;
    pos += n;
    const int w = ibits == 3 ? kW3[idx] : kW4[idx];
    for (int c = 0; c < 3; ++c) {
      const int v = (e[region][0][c] * (64 - w) + e[region][1][c] * w + 32) >> 6;
      if (!isSigned) out[i][c] = (uint16_t)((v * 31) >> 6);
      else { const int m = v < 0 ? ((-v) * 31) >> 5 : (v * 31) >> 5; out[i][c] = (uint16_t)(v < 0 ? (0x8000 | m) : m); }
    }
  }
}
}  // namespace bc6h

struct EnvMap {
  uint32_t size = 0, mips = 0;
  // texels[mip][face] -> size_mip*size_mip RGBA16F (4 x uint16, alpha = 1.0)
  std::vector<std::vector<uint16_t>> level;   // level[mip*6+face]
};

// Decode a BC6H_UF16 mip (w x h texels, ceil(w/4) x ceil(h/4) blocks) to RGBA16F.
static inline void bc6h_decode_image(const uint8_t* blocks, uint32_t w, uint32_t h, uint16_t* rgba, bool isSigned = false) {
  const uint32_t bw = (w + 3) / 4, bh = (h + 3) / 4;
  for (uint32_t by = 0; by < bh; ++by) for (uint32_t bx = 0; bx < bw; ++bx) {
    uint16_t px[16][3];
    bc6h::decodeBlock(blocks + 16 * ((size_t)by * bw + bx), px, isSigned);
    for (uint32_t y = 0; y < 4; ++y) for (uint32_t x = 0; x < 4; ++x) {
      const uint32_t X = bx * 4 + x, Y = by * 4 + y;
      if (X >= w || Y >= h) continue;
      uint16_t* d = rgba + 4 * ((size_t)Y * w + X);
      d[0] = px[y * 4 + x][0]; d[1] = px[y * 4 + x][1]; d[2] = px[y * 4 + x][2]; d[3] = 0x3C00;
    }
  }
}

// DDS container (DX10 header, cube, BC6H_UF16 / RGBA16F / RGBA32F), faces +X,-X,+Y,-Y,+Z,-Z,
// face-major with the full mip chain per face.
static inline bool dds_load_cube(const char* path, EnvMap& env, char* err, size_t errLen);

}  // namespace orc

#include "orc_formats.h"
namespace orc {
static inline bool dds_load_cube(const char* path, EnvMap& env, char* err, size_t errLen) {
  FILE* f = fopen(path, "rb");
  if (!f) { snprintf(err, errLen, "cannot open %s", path); return false; }
  std::vector<uint8_t> d;
  { fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); d.resize((size_t)n);
    if (fread(d.data(), 1, (size_t)n, f) != (size_t)n) { fclose(f); snprintf(err, errLen, "short read"); return false; } fclose(f); }
  auto u32 = [&](size_t o) { uint32_t v; std::memcpy(&v, &d[o], 4); return v; };
  if (d.size() < 148 || std::memcmp(d.data(), "DDS ", 4) != 0 || u32(4) != 124) { snprintf(err, errLen, "not a DDS file"); return false; }
  const uint32_t height = u32(12), width = u32(16);
  uint32_t mips = u32(28); if (mips == 0) mips = 1;
  const uint32_t fourCC = u32(84);
  if (fourCC != 0x30315844u) { snprintf(err, errLen, "only DX10-header DDS supported"); return false; }
  const uint32_t dxgi = u32(128), misc = u32(136);
  if (!(misc & 4u) || width != height) { snprintf(err, errLen, "not a cube map"); return false; }
  size_t off = 148;
  env.size = width; env.mips = mips; env.level.assign((size_t)mips * 6, {});
  for (uint32_t face = 0; face < 6; ++face) for (uint32_t m = 0; m < mips; ++m) {
    const uint32_t s = width >> m ? width >> m : 1;
    std::vector<uint16_t>& img = env.level[(size_t)m * 6 + face];
    img.resize((size_t)s * s * 4);
    size_t bytes;
    if (dxgi == 95 || dxgi == 96) { bytes = (size_t)((s + 3) / 4) * ((s + 3) / 4) * 16; if (off + bytes > d.size()) { snprintf(err, errLen, "truncated"); return false; } bc6h_decode_image(&d[off], s, s, img.data(), dxgi == 96); }
    else if (dxgi == 10) { bytes = (size_t)s * s * 8; if (off + bytes > d.size()) { snprintf(err, errLen, "truncated"); return false; } std::memcpy(img.data(), &d[off], bytes); }
    else if (dxgi == 2) { bytes = (size_t)s * s * 16; if (off + bytes > d.size()) { snprintf(err, errLen, "truncated"); return false; }
      for (size_t i = 0; i < (size_t)s * s * 4; ++i) { float v; std::memcpy(&v, &d[off + 4 * i], 4); img[i] = f32_to_f16(v); } }
    else { snprintf(err, errLen, "unsupported DXGI format %u", dxgi); return false; }
    off += bytes;
  }
  return true;
}
}  // namespace orc
