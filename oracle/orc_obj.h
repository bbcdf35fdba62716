// ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked, imported or called by the product path.
// Restatement of XUSG::ObjLoader::Import (RayTracedGGX/XUSG/Optional/XUSGObjLoader.cpp:18-40,
// 72-431) as called by RayTracer::Init (RayTracer.cpp:83-86): Import(file, needNorm=true,
// needAABB=true, forDX=true, swapYZ=false).
// The reference scans the file with fscanf; this restatement tokenises the same stream and
// reproduces the same records:
//   * "v x y z": position, z negated (forDX)                                  (:188-199)
//   * "vn x y z": normal, z negated                                           (:201-213)
//   * "f a b c d ...": fan triangulation (a,b,c),(a,c,d)...; indices 1-based,
//     negative indices count from the END of the full vertex list              (:230-298)
//   * any other record: rest of the line ignored                               (:219-221)
//   * file has vn: per-vertex normal = normalised vn, vertices split when one
//     position is used with two different vn                                   (:300-335)
//   * file has no vn: face normals (e1 x e2, e2 = v2 - v1) normalised and
//     accumulated unweighted per vertex, then normalised                      (:337-384)
//   * finally the whole index array is reversed (forDX && !swapYZ)             (:227)
// Pinned by tests/golden/obj_import.json.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>
#include <algorithm>

namespace orc {

struct ObjMesh {
  std::vector<float> verts;     // 6 floats per vertex: Pos, Nrm (stride 24)
  std::vector<uint32_t> indices;
  float aabbMin[3], aabbMax[3];
};

namespace objdetail {
struct Tokens {
  std::string data; size_t p = 0;
  void skipws() { while (p < data.size() && isspace((unsigned char)data[p])) ++p; }
  bool next(std::string& t) {               // fscanf("%s")
    skipws(); if (p >= data.size()) return false;
    size_t s = p; while (p < data.size() && !isspace((unsigned char)data[p])) ++p;
    t.assign(data, s, p - s); return true;
  }
  void restOfLine() { while (p < data.size() && data[p] != '\n') ++p; if (p < data.size()) ++p; }   // fgets
  bool readInt(long long& v) {              // fscanf("%lld"): skips white space, fails on non-numeric
    skipws(); size_t s = p;
    if (s < data.size() && (data[s] == '-' || data[s] == '+')) ++s;
    if (s >= data.size() || !isdigit((unsigned char)data[s])) return false;
    char* end; v = strtoll(data.c_str() + p, &end, 10); p = (size_t)(end - data.c_str()); return true;
  }
  bool readFloat(float& v) {                // fscanf("%f")
    skipws(); if (p >= data.size()) return false;
    char* end; v = strtof(data.c_str() + p, &end);
    if (end == data.c_str() + p) return false;
    p = (size_t)(end - data.c_str()); return true;
  }
  bool lit(char c) { if (p < data.size() && data[p] == c) { ++p; return true; } return false; }  // literal, no ws skip
};
}  // namespace objdetail

static inline bool obj_import(const char* path, ObjMesh& out) {
  using namespace objdetail;
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  Tokens tk;
  { fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); tk.data.resize((size_t)n);
    if (n > 0 && fread(&tk.data[0], 1, (size_t)n, f) != (size_t)n) { fclose(f); return false; } fclose(f); }

  // First pass: counts (XUSGObjLoader.cpp:72-164)
  uint32_t numVert = 0, numTexc = 0, numNorm = 0;
  { std::string t;
    while (tk.next(t)) {
      if (t[0] == 'v' && t.size() == 1) { ++numVert; tk.restOfLine(); }
      else if (t[0] == 'v' && t.size() > 1 && t[1] == 't') { ++numTexc; tk.restOfLine(); }
      else if (t[0] == 'v' && t.size() > 1 && t[1] == 'n') { ++numNorm; tk.restOfLine(); }
      else if (t[0] == 'v') { /* :146 default: nothing consumed */ }
      else tk.restOfLine();   // faces are counted in the second pass here; the line is skipped
    }
  }
  tk.p = 0;

  std::vector<float> pos(3 * (size_t)numVert, 0.0f);
  std::vector<float> nrm(3 * (size_t)numVert, 0.0f);
  std::vector<float> fileNormals; fileNormals.reserve(3 * (size_t)numNorm);
  std::vector<uint32_t> idx, nIdx;
  uint32_t vcount = 0;
  std::string t;
  while (tk.next(t)) {
    if (t[0] == 'f') {                                    // loadIndices (:230-298)
      uint32_t v[3] = {0, 0, 0}, vn[3] = {0, 0, 0};
      auto readRef = [&](uint32_t& vi_out, uint32_t& vn_out) -> bool {
        long long vi;
        if (!tk.readInt(vi)) return false;
        vi_out = (uint32_t)(vi < 0 ? vi + (long long)numVert : vi - 1);
        if (numTexc) { if (tk.lit('/')) { long long ti; tk.readInt(ti); } }
        else if (numNorm) tk.lit('/');
        if (numNorm) { if (tk.lit('/')) { long long ni; if (tk.readInt(ni)) vn_out = (uint32_t)(ni < 0 ? ni + (long long)numNorm : ni - 1); } }
        return true;
      };
      bool ok = true;
      for (int i = 0; i < 3 && ok; ++i) ok = readRef(v[i], vn[i]);
      if (!ok) continue;
      for (int i = 0; i < 3; ++i) { idx.push_back(v[i]); if (numNorm) nIdx.push_back(vn[i]); }
      v[1] = v[2]; vn[1] = vn[2];
      while (readRef(v[2], vn[2])) {
        idx.push_back(v[0]); idx.push_back(v[1]); idx.push_back(v[2]);
        if (numNorm) { nIdx.push_back(vn[0]); nIdx.push_back(vn[1]); nIdx.push_back(vn[2]); }
        v[1] = v[2]; vn[1] = vn[2];
      }
    } else if (t[0] == 'v' && t.size() == 1) {
      float x = 0, y = 0, z = 0; tk.readFloat(x); tk.readFloat(y); tk.readFloat(z);
      if (vcount < numVert) { pos[3 * vcount] = x; pos[3 * vcount + 1] = y; pos[3 * vcount + 2] = -z; }
      ++vcount;
    } else if (t[0] == 'v' && t.size() > 1 && t[1] == 'n') {
      float x = 0, y = 0, z = 0; tk.readFloat(x); tk.readFloat(y); tk.readFloat(z);
      fileNormals.push_back(x); fileNormals.push_back(y); fileNormals.push_back(-z);
    } else if (t[0] == 'v') { /* vt and others: not consumed here (:185-216) */ }
    else tk.restOfLine();
  }

  // computePerVertexNormals (:300-335)
  if (!fileNormals.empty()) {
    std::vector<uint32_t> vni(numVert, 0xFFFFFFFFu);
    for (size_t i = 0; i < idx.size(); ++i) {
      uint32_t vi = idx[i];
      if (vni[vi] == nIdx[i]) continue;
      if (vni[vi] < 0xFFFFFFFFu) {                         // split vertex
        const uint32_t src = idx[i];
        vi = (uint32_t)(pos.size() / 3);
        for (int k = 0; k < 3; ++k) { pos.push_back(pos[3 * src + k]); nrm.push_back(nrm[3 * src + k]); }
        idx[i] = vi;
      } else vni[vi] = nIdx[i];
      float n[3] = {fileNormals[3 * nIdx[i]], fileNormals[3 * nIdx[i] + 1], fileNormals[3 * nIdx[i] + 2]};
      const float l = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
      for (int k = 0; k < 3; ++k) nrm[3 * vi + k] = n[k] / l;
    }
  }
  // reverse (:227)
  std::reverse(idx.begin(), idx.end());
  // recomputeNormals (:337-384) when the file has no vn
  if (fileNormals.empty()) {
    const size_t numTri = idx.size() / 3;
    for (size_t i = 0; i < numTri; ++i) {
      const float* p0 = &pos[3 * idx[3 * i]]; const float* p1 = &pos[3 * idx[3 * i + 1]]; const float* p2 = &pos[3 * idx[3 * i + 2]];
      const float e1[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
      const float e2[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
      float n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
      const float l = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
      for (int k = 0; k < 3; ++k) n[k] /= l;
      for (int c = 0; c < 3; ++c) for (int k = 0; k < 3; ++k) nrm[3 * idx[3 * i + c] + k] += n[k];
    }
    const size_t nv = pos.size() / 3;
    for (size_t i = 0; i < nv; ++i) {
      float* n = &nrm[3 * i];
      const float l = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
      for (int k = 0; k < 3; ++k) n[k] /= l;
    }
  }
  const size_t nv = pos.size() / 3;
  out.verts.resize(6 * nv);
  for (size_t i = 0; i < nv; ++i) for (int k = 0; k < 3; ++k) { out.verts[6 * i + k] = pos[3 * i + k]; out.verts[6 * i + 3 + k] = nrm[3 * i + k]; }
  out.indices = idx;
  // computeAABB (:386-416)
  for (int k = 0; k < 3; ++k) out.aabbMin[k] = out.aabbMax[k] = nv ? pos[k] : 0.0f;
  for (size_t i = 1; i < nv; ++i) for (int k = 0; k < 3; ++k) {
    const float x = pos[3 * i + k];
    if (x < out.aabbMin[k]) out.aabbMin[k] = x; else if (x > out.aabbMax[k]) out.aabbMax[k] = x;
  }
  return true;
}

}  // namespace orc
