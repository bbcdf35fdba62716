// ORACLE / TEST INFRASTRUCTURE ONLY -- never linked into, loaded by or shipped with the product.
//
// C entry points over the REFERENCE's own OBJ importer, XUSG::ObjLoader, compiled from where it lies:
//   /root/reference/RayTracedGGX/XUSG/Optional/XUSGObjLoader.cpp (+ XUSGObjLoader.h)
// by `make -C oracle ref` (build container only: /root/reference does not exist on the GPU box; the built
// oracle/_ref/libobjloader_ref.so travels there like the product's own .so files, and is git-ignored).
// This file holds no reference code: it includes the reference's header and calls its public methods
// (ObjLoader::Import / GetNumVertices / GetNumIndices / GetVertices / GetIndices / GetAABB, XUSGObjLoader.h:32-43).
//
// How the reference source is compiled (all on the g++ command line of oracle/Makefile, nothing written in its place):
//   * the standard headers its precompiled header would have supplied are force-included (-include cstdio ... vector);
//   * the three C11 Annex-K names it uses -- fopen_s, fscanf_s, sscanf_s, which glibc does not provide -- are mapped by -D to
//     fopen / fscanf / sscanf (fscanf_s's extra buffer-size argument for "%s" is then an ignored surplus argument).
// It is the only file of the hot path that compiles without Windows/D3D12/XUSG facilities.
// Used by tests/test_obj_reference.py (product importer == reference importer, array for array) and by
// tests/golden/make_obj_golden.py (the known-answer hashes of tests/golden/obj_import.json).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include "XUSGObjLoader.h"

static XUSG::ObjLoader* g_loader = nullptr;

extern "C" {

// ObjLoader::Import(file, needNorm = true, needAABB = true) as RayTracer::Init calls it (RayTracer.cpp:83-84).
int ref_obj_import(const char* path, uint32_t* numVerts, uint32_t* numIndices, uint32_t* stride, float* aabb6) {
  delete g_loader;
  g_loader = new XUSG::ObjLoader();
  if (!g_loader->Import(path, true, true)) { delete g_loader; g_loader = nullptr; return -1; }
  *numVerts = g_loader->GetNumVertices(); *numIndices = g_loader->GetNumIndices(); *stride = g_loader->GetVertexStride();
  if (aabb6) { const XUSG::ObjLoader::AABB& a = g_loader->GetAABB(); const float v[6] = {a.Min.x, a.Min.y, a.Min.z, a.Max.x, a.Max.y, a.Max.z}; std::memcpy(aabb6, v, sizeof v); }
  return 0;
}
int ref_obj_copy(void* verts, uint32_t* indices) {
  if (!g_loader) return -1;
  std::memcpy(verts, g_loader->GetVertices(), (size_t)g_loader->GetNumVertices() * g_loader->GetVertexStride());
  std::memcpy(indices, g_loader->GetIndices(), (size_t)g_loader->GetNumIndices() * sizeof(uint32_t));
  return 0;
}

}  // extern "C"
