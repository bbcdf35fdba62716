#include "ObjLoader.h"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace {

// Character cursor over the whole file; records are white-space separated tokens exactly as a
// scanf("%s") / scanf("%f") / scanf("%lld") consumer would see them.
struct Cursor {
  const char* p; const char* end;
  void skipSpace() { while (p < end && std::isspace((unsigned char)*p)) ++p; }
  void skipLine() { while (p < end && *p != '\n') ++p; if (p < end) ++p; }
  bool word(std::string& out) { skipSpace(); if (p >= end) return false; const char* b = p; while (p < end && !std::isspace((unsigned char)*p)) ++p; out.assign(b, p); return true; }
  bool real(float& v) { skipSpace(); if (p >= end) return false; char* e; v = std::strtof(p, &e); if (e == p) return false; p = e; return true; }
  bool integer(long long& v) {
    skipSpace();
    const char* q = p;
    if (q < end && (*q == '-' || *q == '+')) ++q;
    if (q >= end || !std::isdigit((unsigned char)*q)) return false;
    char* e; v = std::strtoll(p, &e, 10); p = e; return true;
  }
  bool slash() { if (p < end && *p == '/') { ++p; return true; } return false; }
};

struct Corner { long long v, vn; };

}  // namespace

bool ObjLoader::Import(const char* pszFilename, bool needNorm, bool needAABB, bool forDX, bool swapYZ) {
  (void)needNorm;
  FILE* f = std::fopen(pszFilename, "rb");
  if (!f) return false;
  std::string text;
  { char buf[1 << 16]; size_t n; while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, n); }
  std::fclose(f);

  std::vector<float3> positions, normals;
  std::vector<Corner> corners;                 // three per triangle, file order
  bool hasTexc = false, hasNorm = false;
  {  // the reference decides how to read "a/b/c" from whether the FILE has vt / vn records at all
    Cursor c{text.data(), text.data() + text.size()};
    std::string w;
    while (c.word(w)) {
      if (w == "vt") hasTexc = true; else if (w == "vn") hasNorm = true;
      if (!(w[0] == 'v' && w.size() > 1 && w[1] != 't' && w[1] != 'n')) c.skipLine();
    }
  }
  Cursor c{text.data(), text.data() + text.size()};
  std::string w;
  auto corner = [&](Corner& out) -> bool {
    if (!c.integer(out.v)) return false;
    out.vn = 0;
    if (hasTexc) { if (c.slash()) { long long t; c.integer(t); } }
    else if (hasNorm) c.slash();
    if (hasNorm && c.slash()) { long long n; if (c.integer(n)) out.vn = n; }
    return true;
  };
  while (c.word(w)) {
    if (w == "v") {
      float3 p{0, 0, 0}; c.real(p.x); c.real(p.y); c.real(p.z);
      if (swapYZ) std::swap(p.y, p.z);
      if (forDX) p.z = -p.z;
      positions.push_back(p);
    } else if (w == "vn") {
      float3 n{0, 0, 0}; c.real(n.x); c.real(n.y); c.real(n.z);
      if (swapYZ) std::swap(n.y, n.z);
      if (forDX) n.z = -n.z;
      normals.push_back(n);
    } else if (w[0] == 'f') {
      Corner a, b, d;
      if (!corner(a) || !corner(b) || !corner(d)) continue;
      corners.push_back(a); corners.push_back(b); corners.push_back(d);
      b = d;
      while (corner(d)) { corners.push_back(a); corners.push_back(b); corners.push_back(d); b = d; }   // triangle fan
    } else if (w[0] == 'v') {
      // vt and unknown v-records: the following numbers are consumed as unknown records below
    } else c.skipLine();
  }

  const long long numPos = (long long)positions.size(), numNrm = (long long)normals.size();
  std::vector<float3> outPos(positions), outNrm(positions.size(), float3{0, 0, 0});
  m_indices.resize(corners.size());
  std::vector<uint32_t> nIdx(hasNorm ? corners.size() : 0);
  for (size_t i = 0; i < corners.size(); ++i) {
    m_indices[i] = (uint32_t)(corners[i].v < 0 ? corners[i].v + numPos : corners[i].v - 1);
    if (hasNorm) nIdx[i] = (uint32_t)(corners[i].vn < 0 ? corners[i].vn + numNrm : corners[i].vn - 1);
  }

  if (!normals.empty()) {
    // per-vertex normals from the file; a position used with two different normals is duplicated
    std::vector<uint32_t> assigned(positions.size(), 0xFFFFFFFFu);
    for (size_t i = 0; i < m_indices.size(); ++i) {
      uint32_t vi = m_indices[i];
      if (assigned[vi] == nIdx[i]) continue;
      if (assigned[vi] != 0xFFFFFFFFu) {
        outPos.push_back(outPos[vi]); outNrm.push_back(outNrm[vi]);
        vi = (uint32_t)(outPos.size() - 1);
        m_indices[i] = vi;
      } else assigned[vi] = nIdx[i];
      float3 n = normals[nIdx[i]];
      const float len = std::sqrt(n.x * n.x + n.y * n.y + n.z * n.z);
      outNrm[vi] = float3{n.x / len, n.y / len, n.z / len};
    }
  }
  if ((forDX && !swapYZ) || (!forDX && swapYZ)) std::reverse(m_indices.begin(), m_indices.end());
  if (normals.empty()) {
    // face normals, accumulated unweighted, then normalised
    for (size_t t = 0; t + 2 < m_indices.size(); t += 3) {
      const float3 &a = outPos[m_indices[t]], &b = outPos[m_indices[t + 1]], &d = outPos[m_indices[t + 2]];
      const float3 e1{b.x - a.x, b.y - a.y, b.z - a.z}, e2{d.x - b.x, d.y - b.y, d.z - b.z};
      float3 n{e1.y * e2.z - e1.z * e2.y, e1.z * e2.x - e1.x * e2.z, e1.x * e2.y - e1.y * e2.x};
      const float len = std::sqrt(n.x * n.x + n.y * n.y + n.z * n.z);
      n.x /= len; n.y /= len; n.z /= len;
      for (int k = 0; k < 3; ++k) { float3& acc = outNrm[m_indices[t + k]]; acc.x += n.x; acc.y += n.y; acc.z += n.z; }
    }
    for (float3& n : outNrm) { const float len = std::sqrt(n.x * n.x + n.y * n.y + n.z * n.z); n.x /= len; n.y /= len; n.z /= len; }
  }
  m_vertices.resize(outPos.size() * 6);
  for (size_t i = 0; i < outPos.size(); ++i) {
    float* v = &m_vertices[6 * i];
    v[0] = outPos[i].x; v[1] = outPos[i].y; v[2] = outPos[i].z; v[3] = outNrm[i].x; v[4] = outNrm[i].y; v[5] = outNrm[i].z;
  }
  if (needAABB && !outPos.empty()) {
    m_aabb.Min = m_aabb.Max = outPos[0];
    for (size_t i = 1; i < outPos.size(); ++i) {
      const float3& p = outPos[i];
      if (p.x < m_aabb.Min.x) m_aabb.Min.x = p.x; else if (p.x > m_aabb.Max.x) m_aabb.Max.x = p.x;
      if (p.y < m_aabb.Min.y) m_aabb.Min.y = p.y; else if (p.y > m_aabb.Max.y) m_aabb.Max.y = p.y;
      if (p.z < m_aabb.Min.z) m_aabb.Min.z = p.z; else if (p.z > m_aabb.Max.z) m_aabb.Max.z = p.z;
    }
  }
  return true;
}
