// Wavefront OBJ import with the observable behaviour of XUSG::ObjLoader::Import as RayTracer::Init
// calls it (RayTracedGGX/Content/RayTracer.cpp:83-86; semantics in
// RayTracedGGX/XUSG/Optional/XUSGObjLoader.cpp:18-40, 72-431): needNorm, needAABB, forDX, !swapYZ.
// Output: vertices {float3 Pos; float3 Nrm} at stride 24, 32-bit indices; primitive k of the file
// becomes primitive T-1-k with reversed winding (z is negated and the whole index array reversed).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

class ObjLoader {
 public:
  struct float3 { float x, y, z; };
  struct AABB { float3 Min, Max; };

  bool Import(const char* pszFilename, bool needNorm = true, bool needAABB = true, bool forDX = true, bool swapYZ = false);

  uint32_t GetNumVertices() const { return (uint32_t)(m_vertices.size() / 6); }
  uint32_t GetNumIndices() const { return (uint32_t)m_indices.size(); }
  uint32_t GetVertexStride() const { return 24; }
  const uint8_t* GetVertices() const { return reinterpret_cast<const uint8_t*>(m_vertices.data()); }
  const uint32_t* GetIndices() const { return m_indices.data(); }
  const AABB& GetAABB() const { return m_aabb; }

 private:
  std::vector<float> m_vertices;   // 6 floats per vertex
  std::vector<uint32_t> m_indices;
  AABB m_aabb{};
};
