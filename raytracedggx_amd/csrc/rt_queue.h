// Ray records / hit keys exchanged between the three kernels of the ray-tracing pass (raytrace.hip, trace.hip).
//
// No atomics anywhere on this path: every wave of rayGenKernel owns one BIN of RT_BIN ray slots (its 8x8 pixel
// sub-tile, at most one reflection and one diffuse ray per pixel), compacts its rays into the front of the bin
// with __ballot/popcount and records how many there are.  Each wave of the trace kernel owns one bin, so no
// queue head is ever contended (device-scope atomics on a shared head word cost ~11 ns each at the memory side
// and serialised the first, work-queue versions of this pass).  The bins exist twice (rtggx_context.h): stream B
// fills and traces one set while the main stream still shades the set of the frame before (RT_SETS sets, rtggx_context.h).
#pragma once
#include "rtggx_context.h"

namespace rt {

// Ray slots per bin (48-byte ray record + 8-byte hit key per slot): a bin is a wave's 8x8 pixel sub-tile, at most one reflection ray per
// pixel -- and one diffuse ray where a material's metallic is below 1.  The context allocates RT_BIN_MIN slots per bin while both
// materials are fully metallic (the sample's default) and grows to RT_BIN with the first frame whose constants carry a metallic below 1
// (capi.hip rtggx_update_frame; round 3: the bins are the largest allocation of an input set, and three kernels stride over them).
#define RT_BIN 128u
#define RT_BIN_MIN 64u

// Adaptive split of the trace launch (trace.hip): what a bin cost in the previous frame, in lane-steps, decides where and
// how it is traced in this one.  Tuned on the 1080p bunny frame (mean bin: ~700 lane-steps, 14 steps of a wave):
#define RT_SPLIT_CAP 16384u  // entries of the split list
#ifndef RT_VIS_STREAM_RAYS
#define RT_VIS_STREAM_RAYS 800000u   // below this many rays per frame the visibility pass runs on its own stream (capi.hip: all-metal frames; measured: -4.6 % at 0.52 M rays, +0.8 % at 2 M)
#endif
// Round 2 (three-stage pipeline, profiles/r02_c_ab_pipeline.txt block 5): with the traversal no longer the frame's longest chain,
// what pays is STARTING the dearer bins first (a lower threshold: 1000 -> 300, the mean bin costs ~700) and splitting fewer of them
// (2400 -> 3200): every extra wave of a split bin is wave-slot time the other two stages want.  The trace kernel alone gets slower
// (0.120 -> 0.134 ms), the frame faster (0.2207 -> 0.2119 ms; dragon -5 %, 4K -3 %).
#ifndef RT_SPLIT_FRONT
#define RT_WIDE_RAYS 200000u     // launches with fewer rays than this: two traversals in flight (capi.hip rtggx_ray_trace)
#define RT_TINY_RAYS 40000u      // ... and with fewer than this: single-wave workgroups, one per item (trace.hip launchTrace)
#define RT_SPLIT_FRONT 300u  // above this a bin goes on the list: the launch starts with the listed bins
#endif
#ifndef RT_SPLIT_WORK
#define RT_SPLIT_WORK 3200u  // above this (per wave) a listed bin gets twice the waves ...
#endif
#ifndef RT_SPLIT_MAX_SHIFT
#define RT_SPLIT_MAX_SHIFT 1u   // ... up to 2^this
#endif

// 48-byte ray record (three 16-byte words; the traversal reads the first two).  Every ray of a frame has the shader's interval
// (TMin, TMax) = (1e-5, 10000) (RayTracing.hlsl:183-198 with the literals of :441,:504): it is not stored.  Rays handed in through
// rtggx_trace_rays bring their own interval in a side array (TraceArgs::tRange).  Round 1's record was 64 bytes: written by ray
// generation, read by the traversal and by shading, a sixth of the frame's memory traffic.
#define RT_RAY_TMIN 1e-5f
#define RT_RAY_TMAX 10000.0f
struct __attribute__((aligned(16))) RayRec {
  float ox, oy, oz, dx;
  float dy, dz; uint32_t pixel, skip /* (inst<<24)|prim the ray starts on, ~0 none */;
  float wx, wy, wz;   // BRDF weight applied to the returned radiance
  uint32_t flags;     // bit0: diffuse hit group
};
// 8-byte hit key of a ray slot: (bits of t << 32) | id, id = (inst<<24)|prim, ~0 = miss.  t > 0, so keys order like
// (t, id): the closest-hit rule (smaller t, ties to the smaller id) is a 64-bit unsigned min, and every job that
// walks part of a ray's tree merges what it found with one fire-and-forget atomicMin.  rayGenKernel initialises the
// key to (TMax, miss); barycentrics are re-derived from (ray, triangle) by whoever needs them (rt_traverse.h).
typedef unsigned long long HitKey;
RT_HD HitKey hitKey(float t, uint32_t id) { union { float f; uint32_t u; } c; c.f = t; return ((HitKey)c.u << 32) | id; }
RT_HD float hitKeyT(HitKey k) { union { float f; uint32_t u; } c; c.u = (uint32_t)(k >> 32); return c.f; }
RT_HD uint32_t hitKeyId(HitKey k) { return (uint32_t)k; }

// Traversal of the rays in bins [0, numBins) (trace.hip).  countRays: add the rays of rows [fp.rowBegin, fp.rowEnd)
// to the per-frame counters.  tilesX x tilesY: the tile grid the bins come from (4 bins per tile), 0 for a plain list.
// sliceShift: waves per bin for the whole launch (chooseSliceShift).  splitCap >= 0 (with sliceShift 0): adaptive split --
// the bins rayGenKernel marked (bits 8.. of binCount, c->splitList with room for splitCap waves) are traced by several
// waves, and the kernel records what each bin cost (c->binWork) for the next frame's decision; -1: off.
uint32_t chooseSliceShift(rtggx_context* c, bool countRays, uint32_t numBins);
// start / stop (may be null): events attached to the kernel's dispatch (begin of execution / completion).
int launchTrace(rtggx_context* c, const FrameParams& fp, hipStream_t s, uint32_t numBins, bool countRays, uint32_t tilesX, uint32_t tilesY, uint32_t sliceShift, int splitCap,
                hipEvent_t start = nullptr, hipEvent_t stop = nullptr);

}  // namespace rt
