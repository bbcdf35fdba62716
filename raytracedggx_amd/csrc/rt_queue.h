// Ray / hit records exchanged between the three kernels of the ray-tracing pass (raytrace.hip, trace.hip).
//
// No atomics anywhere on this path: every wave of rayGenKernel owns one BIN of RT_BIN ray slots (its 8x8 pixel
// sub-tile, at most one reflection and one diffuse ray per pixel), compacts its rays into the front of the bin
// with __ballot/popcount and records how many there are.  Each wave of the trace kernel owns one bin, so no
// queue head is ever contended (device-scope atomics on a shared head word cost ~11 ns each at the memory side
// and serialised the first, work-queue versions of this pass).  The bins exist twice (rtggx_context.h): stream B
// fills and traces set i while the main stream still shades set i^1.
#pragma once
#include "rtggx_context.h"

namespace rt {

#define RT_BIN 128u          // ray slots per bin = 2 rays x 64 pixels

// 64-byte ray record
struct __attribute__((aligned(16))) RayRec {
  float ox, oy, oz, tmin;
  float dx, dy, dz, tmax;
  uint32_t pixel, skip /* (inst<<24)|prim the ray starts on, ~0 none */, flags /* bit0: diffuse hit group */, pad;
  float wx, wy, wz, wpad;   // BRDF weight applied to the returned radiance
};
// 16-byte hit record
struct __attribute__((aligned(16))) HitRec { float t, b1, b2; uint32_t id; /* (inst<<24)|prim, ~0 = miss */ };

// Traversal of the rays in bins [0, numBins) (trace.hip).  countRays: add the rays of rows [fp.rowBegin, fp.rowEnd)
// to the per-frame counters.  tilesX x tilesY: the tile grid the bins come from (4 bins per tile), 0 for a plain list.
int launchTrace(rtggx_context* c, const FrameParams& fp, hipStream_t s, uint32_t numBins, bool countRays, uint32_t tilesX, uint32_t tilesY);

}  // namespace rt
