// DDS container parse for cube maps: the host half of DDS::Loader::CreateTextureFromFile as used at
// RayTracedGGX/Content/RayTracer.cpp:143-150 (maxsize 8192, forceSRGB=false).  Only the container
// is read here; the texel payload (BC6H_UF16 / BC6H_SF16 blocks, RGBA16F or RGBA32F) is handed to
// rtggx_set_env() untouched and decoded on the device.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace DDS {

struct CubeImage {
  int format = 0;            // DXGI_FORMAT number: 95 BC6H_UF16, 96 BC6H_SF16, 10 R16G16B16A16_FLOAT, 2 R32G32B32A32_FLOAT
  uint32_t size = 0, mips = 0;
  std::vector<uint8_t> payload;   // face-major, full mip chain per face (+X -X +Y -Y +Z -Z)
};

class Loader {
 public:
  bool LoadCubeFromFile(const char* fileName, CubeImage& out, std::string& error) const {
    FILE* f = std::fopen(fileName, "rb");
    if (!f) { error = std::string("cannot open ") + fileName; return false; }
    std::vector<uint8_t> d;
    { uint8_t buf[1 << 16]; size_t n; while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n); }
    std::fclose(f);
    auto u32 = [&](size_t o) { uint32_t v; std::memcpy(&v, &d[o], 4); return v; };
    if (d.size() < 128 || std::memcmp(d.data(), "DDS ", 4) != 0 || u32(4) != 124 || u32(76) != 32) { error = "not a DDS file"; return false; }
    const uint32_t height = u32(12), width = u32(16), caps2 = u32(112);
    uint32_t mips = u32(28); if (!mips) mips = 1;
    const uint32_t pfFlags = u32(80), fourCC = u32(84);
    size_t offset = 128;
    int format = 0; bool cube = (caps2 & 0x200u) != 0;
    if ((pfFlags & 0x4u) && fourCC == 0x30315844u) {          // "DX10"
      if (d.size() < 148) { error = "truncated DX10 header"; return false; }
      format = (int)u32(128); cube = cube || (u32(136) & 0x4u);
      offset = 148;
    } else if ((pfFlags & 0x4u) && fourCC == 113) format = 10;   // D3DFMT_A16B16G16R16F
    else if ((pfFlags & 0x4u) && fourCC == 116) format = 2;      // D3DFMT_A32B32G32R32F
    else { error = "unsupported DDS pixel format"; return false; }
    if (!cube || width != height) { error = "not a cube map"; return false; }
    if (width == 0 || width > 8192) { error = "cube map size out of range (1..8192)"; return false; }
    // the header's mip count is untrusted: at most a full chain (size >> (mips - 1) must still be >= 1)
    if (mips > 14 || (width >> (mips - 1)) == 0) { error = "mip count " + std::to_string(mips) + " exceeds the full chain of a " + std::to_string(width) + " cube"; return false; }
    if (format != 95 && format != 96 && format != 10 && format != 2) { error = "unsupported DXGI format " + std::to_string(format); return false; }
    size_t perFace = 0;
    for (uint32_t m = 0; m < mips; ++m) {
      const uint32_t s = (width >> m) ? (width >> m) : 1;
      perFace += (format == 95 || format == 96) ? (size_t)((s + 3) / 4) * ((s + 3) / 4) * 16 : (size_t)s * s * (format == 10 ? 8 : 16);
    }
    if (d.size() < offset + perFace * 6) { error = "truncated payload"; return false; }
    out.format = format; out.size = width; out.mips = mips;
    out.payload.assign(d.begin() + (long)offset, d.begin() + (long)(offset + perFace * 6));
    return true;
  }
};

}  // namespace DDS
