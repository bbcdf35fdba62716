// Kernel 2 of the ray-tracing pass: TraceRay (RayTracedGGX/Content/Shaders/RayTracing.hlsl:183-198) over the ray
// bins filled by rayGenKernel (raytrace.hip, rt_queue.h).
//
// One wave = one bin: the (usually <= 64) rays of one 8x8 pixel sub-tile, which start next to each other and mostly
// walk the same nodes.  What the measurements settled (tools/microbench, profiles/r01_*):
//   * a traversal step is a dependent gather.  The vector L1 serves a divergent gather at ~1 lane-request (<= 16 B)
//     per cycle per CU whatever the width (tools/microbench/lanecost.hip), L2 answers in ~250 cycles: the cost of a
//     step is the NUMBER of load instructions x active lanes, plus ~300 VALU instructions.  Hence a 4-wide BVH
//     (lbvh.hip, emitNodes4): 128-byte nodes = one cache line = 7 x 16-byte loads, half the steps of the binary tree;
//   * the loads of a step are issued together and pinned in registers before use (hipcc otherwise narrows them per
//     use and sinks them behind branches: several round trips per step);
//   * no persistent threads, no global work queue: rays are short (~10 node visits, ~1.8 triangle tests) and there
//     are only ~2 per resident lane; a persistent kernel that re-dealt rays to idle lanes spent more on dealing
//     (atomics on list heads, refill fetches) than it saved: 0.55 ms against 0.22 ms for one-wave-per-bin on the
//     1080p bunny frame.  The hardware dispatcher balances the ~10^4 short-lived waves over the CUs;
//   * what remains is imbalance INSIDE a wave: see "work sharing" at the kernel;
//   * the world-space ray stays in registers as two object-space rays: switching from the ground instance to the
//     model costs no fetch.
// Per-lane traversal stack in LDS ([entry][lane] layout: conflict-free ds_read/ds_write_b32), RT_STACK entries,
// deeper pushes spill to global memory (launchTrace sizes the spill area from the depth of the built trees; the
// bunny and dragon trees never need it).  Semantics (DESIGN.md "Traversal"): two-level, rays carried into each
// instance's object space, nearer child first, watertight ray/triangle test (Woop, Benthin, Wald 2013), no
// culling, TMin < t < TMax, ties to the lower (instance, primitive).  The closest hit does not depend on the
// order in which boxes are visited, so the 4-wide collapse, the work sharing and the postponed triangle tests
// leave every hit record bit-identical to the oracle's binary-tree walk (tests/test_gpu_parity.py).
#include <type_traits>
#include <hip/hip_ext.h>
#include "rt_queue.h"
#include "rt_traverse.h"

namespace rt {

#define RT_PRIME_LAUNCHES 4u  // launches of a context whose ray / split counters are read back synchronously (launchTrace)
#ifndef RT_STACK
#define RT_STACK 16          // LDS stack entries per lane (deepest stack seen on the bunny/dragon frames: 18, see tools/probes/trace_stats_probe.py; deeper entries spill)
#endif

struct TraceArgs {
  const float4* nodes0; const float4* tris0;   // 128-byte 4-wide nodes (8 x float4), 64-byte leaf triangles (4 x float4)
  const float4* nodes1; const float4* tris1;
  int32_t root0, root1;
  uint32_t haveMesh0, haveMesh1;
  const RayRec* rays; HitKey* hits;
  const float2* tRange;      // (TMin, TMax) per ray slot for rtggx_trace_rays; null: the shader's constants (rt_queue.h)
  const uint32_t* binCount; uint32_t numBins, binSlots;      // slots per bin (rt_queue.h)
  int32_t* overflow;        // [entry][RT_SPILL_WAVES * 64] spill area for stacks deeper than RT_STACK, by resident wave (see launchTrace)
  uint32_t* rayTotals;       // 256 per-frame partial counters (the current frame parity's half)
  uint32_t* stats;           // 768 RT_TRACE_STATS words
  unsigned long long* runTotals;   // 256 running totals (rtggx_ray_total)
  uint32_t countRowBegin, countRowEnd, width;
  size_t spillStride;
  uint32_t sliceShift;       // 0..3: a wave starts with 64, 32, 16, 8 rays of its bin (1, 2, 4, 8 waves per bin; the other lanes start as helpers)
  uint32_t tilesX, tilesY;   // tile grid of the frame (4 bins per 16x16 tile); tilesX == 0: bins are a plain list (rtggx_trace_rays)
  const uint32_t* tileWords; // one word per tile, 0 = nothing drawn there, no rays (rtggx_context.h visDirtyBuf); null with a plain list
  // adaptive split: the first splitBlocks workgroups take their (bin, slice) from the split list; a bin marked as split
  // (binCount bits 8..) is left to them.  binWork: lane-steps spent per bin, for the next frame's decision; null when off.
  const uint32_t* splitList; const uint32_t* splitCount; uint32_t* binWork; uint32_t splitBlocks;
  // the top of both trees for the LDS (rtggx_device.h RT_TOP_*): Bvh4Node records, breadth-first
  const float4* top0; const float4* top1; uint32_t topCount0, topCount1;
  unsigned long long* stamps; uint32_t launch;      // see the kernel
  uint32_t totalItems;       // work items of the launch: what used to be single-wave workgroups, one per (bin, slice); see traceKernel
};

// Wave w of the grid traces the rays of bin w, 64 at a time (a second round only where a sub-tile has more than 64
// rays, i.e. diffuse rays besides the reflection rays).
//
// Work sharing inside the wave.  Ray lengths are very uneven (mean 12 steps, longest ~200): without sharing the
// wave runs at 44% lane utilisation and the kernel ends with a long tail of waves in which one lane chases one ray
// at ~2000 cycles per dependent step.  So a lane is not tied to its ray: the unit of work is a JOB -- a subtree of
// one ray -- and in every step in which at least half of the lanes have no job, those lanes take the BOTTOM stack
// entry (the farthest, largest pending subtree) of lanes that have one.  The helper copies the ray and the victim's
// best hit so far, walks the subtree with its own stack and merges what it finds into the ray's hit key (rt_queue.h)
// with a 64-bit atomic min, which is exactly the closest-hit rule (smaller t, then smaller id) whatever the order in
// which the subtrees finish.  Nobody waits for anybody: the kernel boundary is the only join.
// Tuned on the 1080p bunny frame (tools/sweep: kernel 0.26 ms without sharing, 0.19 ms with these):
#ifndef RT_STEAL_MIN_IDLE
#define RT_STEAL_MIN_IDLE 32u    // share work once half of the lanes have none
#endif
#ifndef RT_STEAL_ROUNDS
#define RT_STEAL_ROUNDS 2        // entries a lane can give away per step
#endif
#ifndef RT_LEAF_BATCH
#define RT_LEAF_BATCH 8u         // lanes standing on a leaf that make a triangle-test phase worthwhile
#endif
// Resident workgroups that draw their bins.  Round 1 launched one single-wave workgroup per bin and let the dispatcher keep the
// chip full of them (24 waves and 150 KB of LDS stacks per CU).  The kernel is now ONE workgroup of RT_TRACE_WAVES waves per CU
// (launchTrace) whose waves draw work ITEMS -- what used to be a workgroup id: a (bin, slice) in dispatch order -- until none are
// left.  Two reasons (profiles/r02_j_resident_trace.txt):
//   * the frame is a pipeline of three streams (capi.hip), and what the trace kernel occupies is not available to the shading and
//     denoising kernels next to it.  Ten resident waves and 55 KB of LDS per CU make the kernel itself 10 % slower than 24 waves
//     and 150 KB did (0.117 against 0.105 ms alone) -- and the frame 7 % faster (0.198 against 0.213 ms).  Six waves are too few
//     (the traversal becomes the longest stage), twelve and more, or more than ~80 KB of LDS, take the gain away again;
//   * a workgroup that lives for the whole launch can afford to keep the top of both trees in LDS (RT_TOP_NODES nodes as 7 x 16
//     bytes, field-major so that lanes on different nodes spread over the banks): 38 % of all node visits on the 1080p bunny frame
//     go to the model's top four levels (85 nodes) and 13 % to the ground's five nodes (profiles/r02_j_levels.txt).  Those visits
//     cost ds_read_b128s instead of requests to the vector L1, the unit the whole frame queues for: worth 2 % on the kernel alone
//     and 5 % on the frame.  A fifth level (341 nodes, 46 KB) adds nothing.
// Items keep their XCD (item & 7 = the XCD of the workgroup id they replace, so the tile -> L2 assignment below stands).
#ifndef RT_PREFETCH
#define RT_PREFETCH 0     // touch-prefetch of the next record: measured +20 % kernel time (one more divergent load per step; profiles/r02_d_limiter.txt)
#endif
#define RT_SPILL_WAVES 16384u    // waves a trace launch may consist of (launchTrace: resident workgroups use 256 x 10-16 of them); sizes the stack spill area
#define RT_TOP_FIELDS 7          // float4 fields of a Bvh4Node the traversal reads
template <int WAVES, int TOP> struct TraceLds {
  int32_t stack[RT_STACK * 64 * WAVES];
  uint32_t victims[64 * WAVES];              // scratch: the lanes offering work, compacted
  uint32_t next, done, pad[2];
  float4 top[TOP ? RT_TOP_FIELDS * RT_TOP_NODES : 1];           // field f of table node k at [f * RT_TOP_NODES + k]; mesh 1's nodes start at k = RT_TOP_SLOT0
};
// LDS pointers carry their address space: a generic pointer that may be LDS or global turns every node fetch into a flat load
typedef __attribute__((address_space(3))) int32_t LdsInt;
typedef __attribute__((address_space(3))) uint32_t LdsUint;
typedef float __attribute__((ext_vector_type(4))) NativeFloat4;      // float4 is a class: it has no copy from another address space
typedef __attribute__((address_space(3))) NativeFloat4 LdsFloat4;
template <int TOP> __device__ __forceinline__ void traceItem(const FrameParams& fp, const TraceArgs& A, LdsInt* const stackMem, LdsUint* const victimMem, const LdsFloat4* const topMem, const uint32_t item, const uint32_t waveSlot) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (item >> 3) & 3u, vblock = ((item >> 5) << 3) | (item & 7u), ldsWave = threadIdx.x >> 6;
  // Workgroup -> tile.  Consecutive workgroup ids go round-robin to the 8 XCDs, each with its own L2; handing every
  // XCD whole super-tiles of 8x8 tiles (128x128 pixels), dealt round-robin over the screen, keeps the part of the tree
  // an L2 needs at any one time small without tying an XCD to one (cheap or expensive) region of the screen.
  // On a small frame or a thin strip the bins are too few to fill the chip, and the kernel lasts as long as its most
  // expensive bin (~70 dependent steps): such launches split every bin over 2, 4 or 8 waves (sliceShift), whose spare
  // lanes start as helpers of the wave's own rays.
  //
  // Adaptive split.  On a full-size frame the kernel still ends with a tail: a few percent of the bins (silhouettes,
  // contact regions) cost 5x the mean, and a wave's steps are dependent.  What a bin cost is known from the previous
  // frame (binWork, lane-steps; the camera moves little in 1/60 s): rayGenKernel gives such bins 2, 4 or 8 waves and
  // lists them (splitList), and the first splitBlocks workgroups of the grid -- dispatched first -- trace them.
  uint32_t shift = A.sliceShift, bin, slice;
  if (vblock < A.splitBlocks) {
    const uint32_t item = vblock * 4u + wave;
    if (item >= min(*A.splitCount, A.splitBlocks * 4u)) return;
    const uint32_t e = A.splitList[item];
    if (e == 0xFFFFFFFFu) return;
    bin = e & 0xFFFFFFu; slice = (e >> 24) & 15u; shift = e >> 28;
    if (bin >= A.numBins) return;
  } else {
    const uint32_t slices = 1u << shift;
    const uint32_t blk = (vblock - A.splitBlocks) >> shift, sub = (vblock - A.splitBlocks) & (slices - 1u);
    uint32_t tile = blk;
    if (A.tilesX != 0u) {
      const uint32_t xcd = blk & 7u, local = blk >> 3;
      const uint32_t super = (local >> 6) * 8u + xcd, inSuper = local & 63u;
      const uint32_t superX = (A.tilesX + 7u) >> 3;
      const uint32_t tx = (super % superX) * 8u + (inSuper & 7u), ty = (super / superX) * 8u + (inSuper >> 3);
      if (tx >= A.tilesX || ty >= A.tilesY) return;
      tile = ty * A.tilesX + tx;
      // three quarters of the bunny frame's items are bins of tiles nothing was drawn in: known from a scalar load, where the bin's count
      // below is a vector load a wave waited for eight times in a row
      uint32_t word;
      asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(word) : "s"(A.tileWords), "s"(tile * 4u) : "memory");
      if (word == 0u) return;
    }
    bin = tile * 4u + ((sub * 4u + wave) >> shift); slice = (sub * 4u + wave) & (slices - 1u);
    if (bin >= A.numBins) return;
  }
  const uint32_t raysPerWave = 64u >> shift;
  const uint32_t countWord = A.binCount[bin];
  if (vblock >= A.splitBlocks && A.binWork != nullptr && (countWord >> 8) != 0u) return;      // traced by the waves of the split list
  const uint32_t count = min(countWord & 0xFFu, A.binSlots);
  if (count <= slice * raysPerWave) return;
  LdsInt* const stackBase = stackMem + ldsWave * (RT_STACK * 64);                         // entry e of lane l at [e * 64 + l]
  LdsInt* const stack = stackBase + lane;
  LdsUint* const victims = victimMem + ldsWave * 64;
  const size_t spillStride = A.spillStride;
  int32_t* const spill = A.overflow + (size_t)waveSlot * 64u + lane;                       // entry e at spill[e * spillStride]: a wave of the launch owns its 64 words of every entry
  const unsigned long long laneLt = (1ull << lane) - 1ull;
  uint32_t nRays = 0, work = 0;
#ifdef RT_TRACE_STATS
  uint32_t stNode = 0, stLeaf = 0, stIter = 0, stDeep = 0, stSteal = 0, stLeafPhase = 0;
  const unsigned long long stT0 = clock64(), stW0 = wall_clock64();
#endif
  for (uint32_t base = 0; base < count; base += 64u) {
  const uint32_t rayIndex = base + slice * raysPerWave + lane;
  const uint32_t slot = bin * A.binSlots + min(rayIndex, A.binSlots - 1u);
  const bool hasRay = lane < raysPerWave && rayIndex < count;

  // ---- the ray (world space) ------------------------------------------------------------------------------------
  const uint32_t raySlot = hasRay ? slot : bin * A.binSlots;
  const float4* rp = reinterpret_cast<const float4*>(A.rays + raySlot);
  const float4 q0 = rp[0], q1 = rp[1];      // origin + dx | dy dz pixel skip
  const float4 ra = make_float4(q0.x, q0.y, q0.z, 0.0f), rb = make_float4(q0.w, q1.x, q1.y, 0.0f);
  const uint4 rc = make_uint4(__float_as_uint(q1.z), __float_as_uint(q1.w), 0u, 0u);
  float tmin0 = RT_RAY_TMIN, bestT = RT_RAY_TMAX;
  if (A.tRange != nullptr) { const float2 tr = A.tRange[raySlot]; tmin0 = tr.x; bestT = tr.y; }
  uint32_t bestId = 0xFFFFFFFFu;
  {
    const uint32_t row = rc.x / A.width;
    if (hasRay && row >= A.countRowBegin && row < A.countRowEnd) ++nRays;
  }

  // ---- traversal of one instance (0: the ground, 1: the model), all rays of the round -------------------------------
  // One loop per instance, one after the other: inside a loop the instance is a compile-time constant (node and triangle
  // bases are scalars, no per-lane selection, and a job never has to swap one object-space ray for the other).  A primary
  // job keeps its best hit in registers from the first loop to the second; helpers merge theirs into the ray's key.
  auto traverse = [&](auto instTag) {
  constexpr uint32_t INST = decltype(instTag)::value;
  if ((INST ? A.haveMesh1 : A.haveMesh0) == 0u) return;
  const float4* const nodes = INST ? A.nodes1 : A.nodes0;
  const float4* const tris = INST ? A.tris1 : A.tris0;
  LaneRay r = toObject(ra.x, ra.y, ra.z, rb.x, rb.y, rb.z, fp.invWorld[INST]);
  float tmin = tmin0;
  uint32_t skip = rc.y;
  int32_t cur = (INST ? A.topCount1 : A.topCount0) ? RT_TOP_FLAG : (INST ? A.root1 : A.root0);      // the root is entry 0 of the table
  int sp = 0, sb = 0;                                // my stack holds entries [sb, sp)
  // job state: the primary job of my own ray (none for a degenerate interval: the key stays a miss)
  bool job = hasRay && bestT > tmin, helper = false;
  uint32_t owner = slot;                             // the ray slot whose key this job's hits go to
  float myT = bestT; uint32_t myId = bestId;         // the best hit of the job in hand (a helper's: its victim's, then its own)

#if RT_PREFETCH
  float pf = 0.0f;      // destination of the touch load that warms the L1 line of the record this lane visits next (see the loop's end)
#endif
  for (;;) {
    const unsigned long long jobMask = __ballot(job);
    if (jobMask == 0ull) break;
    work += (uint32_t)__popcll(jobMask);
    // -- idle lanes take over pending subtrees
    if ((uint32_t)__popcll(~jobMask) >= RT_STEAL_MIN_IDLE) {
      for (int round = 0; round < RT_STEAL_ROUNDS; ++round) {
        const bool offers = job && sb < sp && sb < RT_STACK;                 // bottom entry exists and lives in LDS
        const unsigned long long victimMask = __ballot(offers), idleMask = __ballot(!job);
        if (victimMask == 0ull || idleMask == 0ull) break;
        const uint32_t nv = (uint32_t)__popcll(victimMask), ni = (uint32_t)__popcll(idleMask);
        const uint32_t myV = (uint32_t)__popcll(victimMask & laneLt), myI = (uint32_t)__popcll(idleMask & laneLt);
        if (offers) victims[myV] = lane;
        const bool thief = !job && myI < nv;
        const uint32_t v = thief ? victims[myI] : lane;
        // everything a helper needs, read from the victim's registers
        const int vsb = __shfl(sb, (int)v);
        const float v_ox = __shfl(r.ox, (int)v), v_oy = __shfl(r.oy, (int)v), v_oz = __shfl(r.oz, (int)v);
        const float v_ix = __shfl(r.ix, (int)v), v_iy = __shfl(r.iy, (int)v), v_iz = __shfl(r.iz, (int)v);
        const float v_Sx = __shfl(r.Sx, (int)v), v_Sy = __shfl(r.Sy, (int)v), v_Sz = __shfl(r.Sz, (int)v);
        const int v_kx = __shfl(r.kx, (int)v), v_ky = __shfl(r.ky, (int)v), v_kz = __shfl(r.kz, (int)v);
        const float v_tmin = __shfl(tmin, (int)v), v_bestT = __shfl(myT, (int)v);
        const uint32_t v_bestId = (uint32_t)__shfl((int)myId, (int)v), v_skip = (uint32_t)__shfl((int)skip, (int)v);
        const uint32_t v_owner = (uint32_t)__shfl((int)owner, (int)v);
        if (thief) {
          cur = stackBase[vsb * 64 + (int)v];
          r.ox = v_ox; r.oy = v_oy; r.oz = v_oz; r.ix = v_ix; r.iy = v_iy; r.iz = v_iz; r.Sx = v_Sx; r.Sy = v_Sy; r.Sz = v_Sz;
          r.kx = v_kx; r.ky = v_ky; r.kz = v_kz;
          tmin = v_tmin; myT = v_bestT; myId = v_bestId; skip = v_skip; owner = v_owner;
          sp = sb = 0; job = true; helper = true;
#ifdef RT_TRACE_STATS
          ++stSteal;
#endif
        }
        if (offers && myV < ni) ++sb;                                          // my bottom entry was taken
      }
    }
#ifdef RT_TRACE_STATS
    ++stIter; if (job && cur >= 0) { ++stNode; atomicAdd(&A.stats[512 + INST * 32 + ((cur & RT_TOP_FLAG) ? 31 : min(__float_as_int(nodes[(size_t)cur * 8 + 7].x), 30))], 1u); }
#endif
    bool finished = false;
    // Pop the next entry of my stack; with none left the job is over.  (The LDS part of the stack is read
    // unconditionally: keeping the two address spaces apart keeps the pop a ds_read instead of a flat load.)
    auto popOrFinish = [&]() {
      if (sp > sb) {
        --sp;
        cur = stack[min(sp, RT_STACK - 1) * 64];
        asm volatile("" : "+v"(cur));
        if (sp >= RT_STACK) cur = spill[(size_t)(sp - RT_STACK) * spillStride];
        if (sp == sb) sp = sb = 0;
      } else {
        sp = sb = 0;
        job = false; finished = true;
      }
    };
    // -- node phase: lanes standing on a 4-wide node test its boxes (one 128-byte fetch); lanes standing on a leaf wait
    if (job && cur >= 0) {
      float4 q0, q1, q2, q3, q4, q5, q6;
      if (TOP && (cur & RT_TOP_FLAG)) {      // a node of the table in LDS
        const LdsFloat4* rec = topMem + (INST ? RT_TOP_SLOT0 : 0) + (cur & 0xFFFF);
#define RT_TOP_FIELD(q, f) { const NativeFloat4 v = rec[(f) * RT_TOP_NODES]; q = make_float4(v.x, v.y, v.z, v.w); }
        RT_TOP_FIELD(q0, 0) RT_TOP_FIELD(q1, 1) RT_TOP_FIELD(q2, 2) RT_TOP_FIELD(q3, 3) RT_TOP_FIELD(q4, 4) RT_TOP_FIELD(q5, 5) RT_TOP_FIELD(q6, 6)
#undef RT_TOP_FIELD
      } else {
        const float4* rec = nodes + (size_t)cur * 8;
        q0 = rec[0]; q1 = rec[1]; q2 = rec[2]; q3 = rec[3]; q4 = rec[4]; q5 = rec[5]; q6 = rec[6];
      }
      asm volatile("" : "+v"(q0.x), "+v"(q0.y), "+v"(q0.z), "+v"(q0.w), "+v"(q1.x), "+v"(q1.y), "+v"(q1.z), "+v"(q1.w));
      asm volatile("" : "+v"(q2.x), "+v"(q2.y), "+v"(q2.z), "+v"(q2.w), "+v"(q3.x), "+v"(q3.y), "+v"(q3.z), "+v"(q3.w));
      asm volatile("" : "+v"(q4.x), "+v"(q4.y), "+v"(q4.z), "+v"(q4.w), "+v"(q5.x), "+v"(q5.y), "+v"(q5.z), "+v"(q5.w));
      asm volatile("" : "+v"(q6.x), "+v"(q6.y), "+v"(q6.z), "+v"(q6.w));
      // q0..q5 = minx[4] miny[4] minz[4] maxx[4] maxy[4] maxz[4], q6 = ref[4]
      float t0, t1, t2, t3, tf;
      int32_t c0 = __float_as_int(q6.x), c1 = __float_as_int(q6.y), c2 = __float_as_int(q6.z), c3 = __float_as_int(q6.w);
      const float inf = __builtin_inff();
      slabTest(r, q0.x, q1.x, q2.x, q3.x, q4.x, q5.x, tmin, myT, t0, tf); t0 = (t0 <= tf * 1.0000004f && c0 != RT_BVH4_EMPTY) ? t0 : inf;
      slabTest(r, q0.y, q1.y, q2.y, q3.y, q4.y, q5.y, tmin, myT, t1, tf); t1 = (t1 <= tf * 1.0000004f && c1 != RT_BVH4_EMPTY) ? t1 : inf;
      slabTest(r, q0.z, q1.z, q2.z, q3.z, q4.z, q5.z, tmin, myT, t2, tf); t2 = (t2 <= tf * 1.0000004f && c2 != RT_BVH4_EMPTY) ? t2 : inf;
      slabTest(r, q0.w, q1.w, q2.w, q3.w, q4.w, q5.w, tmin, myT, t3, tf); t3 = (t3 <= tf * 1.0000004f && c3 != RT_BVH4_EMPTY) ? t3 : inf;
      // order the entries by entry distance (misses last); the order only affects how soon far boxes get culled
#define RT_CSWAP(ta, ca, tb, cb) { const bool sw = tb < ta; const float tt = sw ? tb : ta; tb = sw ? ta : tb; ta = tt; const int32_t cc = sw ? cb : ca; cb = sw ? ca : cb; ca = cc; }
      RT_CSWAP(t0, c0, t1, c1) RT_CSWAP(t2, c2, t3, c3) RT_CSWAP(t0, c0, t2, c2) RT_CSWAP(t1, c1, t3, c3) RT_CSWAP(t1, c1, t2, c2)
#undef RT_CSWAP
#define RT_PUSH(v) { if (sp < RT_STACK) stack[sp * 64] = (v); else spill[(size_t)(sp - RT_STACK) * spillStride] = (v); ++sp; }
      if (t0 < inf) {
        if (t3 < inf) RT_PUSH(c3)
        if (t2 < inf) RT_PUSH(c2)
        if (t1 < inf) RT_PUSH(c1)
        cur = c0;
#ifdef RT_TRACE_STATS
        if ((uint32_t)sp > stDeep) stDeep = (uint32_t)sp;
#endif
      } else popOrFinish();
#undef RT_PUSH
    }
    // -- leaf phase: the triangle test is the longest stretch of code, so it runs for many lanes at once: when enough
    //    lanes stand on a leaf, or when no lane has a node left to visit
    {
      const bool atLeaf = job && cur < 0;
      const unsigned long long leafMask = __ballot(atLeaf);
      if (leafMask != 0ull && ((uint32_t)__popcll(leafMask) >= RT_LEAF_BATCH || __ballot(job && cur >= 0) == 0ull)) {
#ifdef RT_TRACE_STATS
        ++stLeafPhase;
#endif
        if (atLeaf) {
          // a leaf is one triangle: three 16-byte words of its 64-byte record (the primitive id sits in the third as well)
          const float4* rec = tris + (size_t)(uint32_t)~cur * 4;
          float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
          asm volatile("" : "+v"(q0.x), "+v"(q0.y), "+v"(q0.z), "+v"(q0.w), "+v"(q1.x), "+v"(q1.y), "+v"(q1.z), "+v"(q1.w));
          asm volatile("" : "+v"(q2.x), "+v"(q2.y));
          // q0 = v0.xyz v1.x | q1 = v1.yz v2.xy | q2 = v2.z prim pad pad
          const uint32_t id = (INST << 24) | __float_as_uint(q2.y);
          if (id != skip) {
            float t, b1, b2;
            if (woopTest(r, q0, q1, q2, t, b1, b2) && t > tmin) {
              const bool closer = t < myT;
              const bool tie = myId != 0xFFFFFFFFu && t == myT && id < myId;
              if (closer || tie) { myT = t; myId = id; }
            }
          }
#ifdef RT_TRACE_STATS
          ++stLeaf;
#endif
          popOrFinish();
        }
      }
    }
#if RT_PREFETCH
    // -- touch the record this lane stands on now: its node (one 128-byte line) or leaf triangle is fetched by the NEXT iteration,
    //    behind that iteration's ballots, work sharing and LDS traffic; asking for the line now turns that fetch's L2 round trip
    //    (500-900 cycles under load) into an L1 hit.  One 4-byte load per lane, its value unused.
    if (job) {
      const float* line = cur >= 0 ? reinterpret_cast<const float*>(nodes + (size_t)cur * 8) : reinterpret_cast<const float*>(tris + (size_t)(uint32_t)~cur * 4);
      pf = __builtin_nontemporal_load(line);
    }
#endif
    // -- a finished helper merges its best hit into the ray's key (one that found nothing closer than what it started
    //    with repeats its victim's candidate: harmless); a finished primary job keeps its own in bestT / bestId
    if (finished) {
      if (helper) { if (myId != 0xFFFFFFFFu) atomicMin(&A.hits[owner], hitKey(myT, myId)); helper = false; }
      else { bestT = myT; bestId = myId; }
    }
  }
  };   // traverse
  traverse(std::integral_constant<uint32_t, 0u>{});
  traverse(std::integral_constant<uint32_t, 1u>{});
  if (hasRay && bestId != 0xFFFFFFFFu) atomicMin(&A.hits[slot], hitKey(bestT, bestId));
  }   // base

  // ray statistics: one fire-and-forget atomic per wave, spread over 256 words
  for (int o = 32; o > 0; o >>= 1) nRays += __shfl_down(nRays, o);
  if (lane == 0 && nRays) { atomicAdd(&A.rayTotals[bin & 255u], nRays); atomicAdd(&A.runTotals[bin & 255u], (unsigned long long)nRays); }
  if (lane == 0 && A.binWork != nullptr) { if (vblock < A.splitBlocks) atomicAdd(&A.binWork[bin], work); else A.binWork[bin] = work; }
#ifdef RT_TRACE_STATS
  for (int o = 32; o > 0; o >>= 1) { stNode += __shfl_down(stNode, o); stLeaf += __shfl_down(stLeaf, o); stSteal += __shfl_down(stSteal, o); stDeep = max(stDeep, (uint32_t)__shfl_down((int)stDeep, o)); }
  if (lane == 0) {   // lane node steps, lane leaf steps, wave iterations, waves, deepest stack, wave lifetime (sum, max), most iterations, steals
    uint32_t* st = A.stats + (bin & 15u) * 16u;      // 16 copies to keep the atomics off one word
    atomicAdd(&st[0], stNode); atomicAdd(&st[1], stLeaf); atomicAdd(&st[2], stIter);
    atomicAdd(&st[3], 1u); atomicMax(&st[4], stDeep);
    const uint32_t life = (uint32_t)((clock64() - stT0) >> 4);
    {   // histograms of wave start / end times, 8 us buckets from the stamp of stampKernel (wall clock: 100 MHz)
      const unsigned long long t0 = *reinterpret_cast<const unsigned long long*>(A.stats + 764);
      const unsigned long long now = wall_clock64();
      const uint32_t be = min((uint32_t)((now - t0) / 800ull), 31u);
      const uint32_t bs = min((uint32_t)((stW0 - t0) / 800ull), 31u);
      atomicAdd(&A.stats[256 + be], 1u); atomicAdd(&A.stats[288 + bs], 1u);
    }
    atomicAdd(&st[5], life >> 6); atomicMax(&st[6], life); atomicMax(&st[7], stIter); atomicAdd(&st[8], stSteal); atomicAdd(&st[9], stLeafPhase);
  }
#endif
}

// WAVES waves per workgroup; PER_SIMD: the waves per SIMD the register allocation has to leave room for (launchTrace's variants)
// TOP = 0: no table in LDS (launchTrace passes topCount0 = topCount1 = 0 to such a variant)
template <int WAVES, int PER_SIMD, int TOP> __global__ void __launch_bounds__(64 * WAVES, PER_SIMD) traceKernel(const FrameParams* __restrict__ fpp, TraceArgs A) {
  __shared__ TraceLds<WAVES, TOP> lds;
  // the tables: Bvh4Node records (8 x float4, the last one padding) -> field-major
  if (TOP) for (uint32_t i = threadIdx.x; i < A.topCount0 * 8u; i += 64u * WAVES) { const uint32_t k = i >> 3, f = i & 7u; if (f < RT_TOP_FIELDS) lds.top[f * RT_TOP_NODES + k] = A.top0[i]; }
  if (TOP) for (uint32_t i = threadIdx.x; i < A.topCount1 * 8u; i += 64u * WAVES) { const uint32_t k = i >> 3, f = i & 7u; if (f < RT_TOP_FIELDS) lds.top[f * RT_TOP_NODES + RT_TOP_SLOT0 + k] = A.top1[i]; }
  if (threadIdx.x == 0) { lds.next = 0u; lds.done = 0u; }
  // Time stamps for the choice of the workgroup size (steerTraceWaves; 100 MHz wall clock).  Three slots in turn: this launch writes slot
  // launch % 3 (start; end = the latest workgroup end), reads the previous launch's and clears the next one's.
  if (blockIdx.x == 0 && threadIdx.x == 0 && A.stamps != nullptr) {
    const unsigned long long now = wall_clock64();
    unsigned long long* const mine = A.stamps + 2u * (A.launch % 3u);
    const unsigned long long* const prev = A.stamps + 2u * ((A.launch + 2u) % 3u);
    unsigned long long* const next = A.stamps + 2u * ((A.launch + 1u) % 3u);
    if (A.launch != 0u && prev[0] != 0ull && prev[1] > prev[0]) A.stamps[6] += prev[1] - prev[0];      // running sum of the durations of the launches before this one
    A.stamps[7] = now;
    mine[0] = now; next[0] = 0ull; next[1] = 0ull;
  }
  __syncthreads();
  // Items are dealt to the workgroups of an XCD round-robin: every workgroup gets a sample of the whole list (the expensive bins
  // of the split list first).  The waves of a workgroup draw from that share through a counter in LDS and leave when it is used up.
  // (One counter per XCD in global memory, drawn from by every wave of the chip, cost 0.75 ms per launch: device-scope atomics on
  // one address are served one at a time at the memory side, ~20 ns each; profiles/r02_j_resident_trace.txt.)
  const uint32_t xcd = blockIdx.x & 7u, groupsPerXcd = gridDim.x >> 3;
  for (;;) {
    uint32_t j = 0;
    if ((threadIdx.x & 63u) == 0u) j = atomicAdd(&lds.next, 1u);
    j = (uint32_t)__builtin_amdgcn_readfirstlane((int)j);
    const uint32_t item = ((blockIdx.x >> 3) + j * groupsPerXcd) * 8u + xcd;
    if (j >= 0x1000000u || item >= A.totalItems) break;
    traceItem<TOP>(*fpp, A, (LdsInt*)lds.stack, (LdsUint*)lds.victims, (const LdsFloat4*)lds.top, item, blockIdx.x * (uint32_t)WAVES + (threadIdx.x >> 6));
  }
  // the last wave of the workgroup to leave stamps the end (no barrier: a wave that is done gives its slot back at once)
  if ((threadIdx.x & 63u) == 0u && A.stamps != nullptr && atomicAdd(&lds.done, 1u) == (uint32_t)WAVES - 1u) atomicMax(A.stamps + 2u * (A.launch % 3u) + 1u, wall_clock64());
}

#ifdef RT_TRACE_STATS
__global__ void stampKernel(uint32_t* stats) { *reinterpret_cast<unsigned long long*>(stats + 764) = wall_clock64(); }
#endif

// How many waves per bin for the whole launch: one when the rays fill the chip (~5000 wave slots x 64 lanes); 2, 4 or
// 8 when they do not (small frames, thin strips of a multi-GPU frame).  The ray count is last frame's, copied back
// asynchronously.
// The size of the traversal's one workgroup per CU.  What the traversal keeps resident is not available to the kernels of the other
// two pipeline stages (capi.hip), so it should be small -- 12 waves are best or within noise of best on seven of eight workloads --
// unless the traversal is what the frame waits for (a large mesh with diffuse rays as well: 0.43 ms per frame with 12 waves, 0.37
// with 14; profiles/r02_j_resident_trace.txt).  The kernel stamps its own start and end and keeps a running sum of its durations;
// with the ray counters (every 16th frame) the host gets, for the launches since the last sample, the PERIOD between launches and
// the SHARE of it the traversal ran.  The share alone does not tell the workloads apart that gain from more waves from those that
// lose (0.92 against 0.88-0.91), so the size is tried: with a share above RT_WAVES_TRY (0.90) two more waves for two samples; they stay if
// the period fell by 3 %, else the old size returns and the next trial waits RT_WAVES_RETRY samples (the workload drifts: a turning
// model changes the period by 30 % over a few hundred frames).  A share below RT_WAVES_SHRINK
// takes two waves away again (down to 12).  Results never depend on any of this.
#define RT_WAVES_TRY 0.90f
#define RT_WAVES_SHRINK 0.70f
#define RT_WAVES_RETRY 16u
static void steerTraceWaves(rtggx_context* c, const unsigned long long* stamps, uint32_t launch) {
  const unsigned long long sum = stamps[0], start = stamps[1];
  const unsigned long long dSum = sum - c->traceStampSum, dStart = start - c->traceStampStart;
  const uint32_t launches = launch - c->traceStampAt, windowBegin = c->traceStampAt;
  const bool usable = c->traceStampStart != 0ull && start > c->traceStampStart && sum >= c->traceStampSum && dSum <= dStart && launches >= 8u && launches <= 64u;
  c->traceStampSum = sum; c->traceStampStart = start; c->traceStampAt = launch;
  if (!usable) { c->traceTrial = 0u; return; }
  const float period = (float)dStart / (float)launches;
  c->traceShare = (float)dSum / (float)dStart;
  static const bool log = getenv("RTGGX_TRACE_LOG") != nullptr;
  if (log) fprintf(stderr, "[rtggx] trace sample at launch %u: %u launches, period %.1f us, share %.3f, waves %u, trial %u (base %.1f us), cooldown %u\n", launch, launches, period * 0.01f, c->traceShare, c->traceWaves, c->traceTrial, c->traceTrialBase * 0.01f, c->traceCooldown);
  if (c->traceWavesForced || c->lastTraceSmall) { c->traceTrial = 0u; return; }      // (two launches in flight: their stamps overlap)
  if (c->traceTrial != 0u) {
    if (++c->traceTrial == 2u) return;                 // the first sample after the change mixes both sizes (frames in flight)
    if (period < 0.97f * c->traceTrialBase) c->traceCooldown = 8u;                       // the new size stays
    else { c->traceWaves -= 2u; c->traceWavesSince = launch; c->traceCooldown = RT_WAVES_RETRY; }
    c->traceTrial = 0u;
    return;
  }
  if (c->traceCooldown) --c->traceCooldown;
  const bool clean = windowBegin >= c->traceWavesSince + 8u;      // every launch of this sample's window ran with the current size (up to 4 frames are in flight)
  if (c->traceShare < RT_WAVES_SHRINK && c->traceWaves > 12u) { c->traceWaves -= 2u; c->traceWavesSince = launch; if (c->traceCooldown < 4u) c->traceCooldown = 4u; }
  else if (c->traceShare > RT_WAVES_TRY && c->traceWaves < 16u && c->traceCooldown == 0u && clean) { c->traceTrialBase = period; c->traceWaves += 2u; c->traceWavesSince = launch; c->traceTrial = 1u; }
}

uint32_t chooseSliceShift(rtggx_context* c, bool countRays, uint32_t numBins) {
  if (countRays && c->rayCountersInFlight && hipEventQuery(c->evRayCounters) == hipSuccess) {
    uint32_t sum = 0; for (int i = 0; i < 256; ++i) sum += c->hostRayCounters[i];
    c->lastFrameRays = sum; c->splitDemand = c->hostRayCounters[256]; c->rayCountersInFlight = false;
    steerTraceWaves(c, reinterpret_cast<const unsigned long long*>(c->hostRayCounters + 258), c->traceSampleLaunch);
  }
  const uint32_t raysGuess = countRays ? c->lastFrameRays : numBins * 40u;
  return raysGuess < 25000u ? 3u : raysGuess < 60000u ? 2u : raysGuess < 110000u ? 1u : 0u;
}

int launchTrace(rtggx_context* c, const FrameParams& fp, hipStream_t s, uint32_t numBins, bool countRays, uint32_t tilesX, uint32_t tilesY, uint32_t sliceShift, int splitCap,
                hipEvent_t start, hipEvent_t stop) {
  TraceArgs T;
  if (numBins == 0) return 0;
  if (numBins > c->numBinsMax) { setError("launchTrace: %u bins exceed the %u allocated", numBins, c->numBinsMax); return -1; }
  const bool have0 = c->mesh[0].tris != nullptr, have1 = c->mesh[1].tris != nullptr;
  // a mesh with one triangle has no internal nodes; an absent mesh has nothing: point those bases at the dummy record
  T.nodes0 = (const float4*)(have0 && c->mesh[0].nodes4 ? (const void*)c->mesh[0].nodes4 : c->dummyRecord);
  T.tris0 = (const float4*)(have0 ? (const void*)c->mesh[0].tris : c->dummyRecord);
  T.nodes1 = (const float4*)(have1 && c->mesh[1].nodes4 ? (const void*)c->mesh[1].nodes4 : c->dummyRecord);
  T.tris1 = (const float4*)(have1 ? (const void*)c->mesh[1].tris : c->dummyRecord);
  T.root0 = c->mesh[0].root; T.root1 = c->mesh[1].root; T.haveMesh0 = have0; T.haveMesh1 = have1;
  T.rays = (const RayRec*)c->rayQueue; T.hits = (HitKey*)c->hitQueue; T.tRange = (const float2*)c->traceRayRange;
  T.binCount = c->binCount; T.numBins = numBins; T.binSlots = c->binSlots;
  // Stacks deeper than the LDS part spill to global memory; the built trees say how deep they can get (a 4-wide node leaves its other
  // entries behind: the build adds them up along every path, BuildResult::stack4).  The spill area belongs to the launch's WAVES, not to the bins
  // (round 3; per bin it was 2 x 20 entries x 32 640 bins x 2 KB = 2.7 GB for the bunny at 1080p, 4.3 GB for the dragon, four times
  // that at 4K -- for an area the bunny and dragon frames touch a handful of times): at most RT_SPILL_WAVES waves per launch, which is
  // what caps the grid of the single-wave variant below.
  const uint32_t deepest = (c->mesh[0].stack4 > c->mesh[1].stack4 ? c->mesh[0].stack4 : c->mesh[1].stack4) + 1u;      // what the builds found (lbvh.hip roots4Kernel)
  if (deepest > RT_STACK + c->spillEntries) {
    RT_HIP(hipDeviceSynchronize());
    if (c->stackOverflow) { RT_HIP(hipFree(c->stackOverflow)); c->stackOverflow = nullptr; }
    c->spillEntries = deepest - RT_STACK;
    RT_HIP(hipMalloc(&c->stackOverflow, (size_t)2 * c->spillEntries * RT_SPILL_WAVES * 64 * 4));      // twice: two traversals can be in flight (traceSpillHalf)
  }
  T.overflow = c->stackOverflow + (size_t)c->traceSpillHalf * c->spillEntries * RT_SPILL_WAVES * 64; T.rayTotals = c->rayCounter32; T.stats = c->rayCounterBuf + 1024; T.runTotals = c->rayCounter + 256;
  T.spillStride = (size_t)RT_SPILL_WAVES * 64;
  T.countRowBegin = countRays ? fp.rowBegin : 0u; T.countRowEnd = countRays ? fp.rowEnd : 0u; T.width = fp.W;
#ifdef RT_TRACE_STATS
  hipLaunchKernelGGL(stampKernel, dim3(1), dim3(1), 0, s, c->rayCounterBuf + 1024);
#endif
  T.tilesX = tilesX; T.tilesY = tilesY; T.tileWords = tilesX ? c->traceTileWords : nullptr;
  T.sliceShift = sliceShift;
  const bool adaptive = splitCap >= 0 && sliceShift == 0u && tilesX != 0u;
  T.splitList = c->splitList; T.splitCount = c->splitCount;
  T.binWork = adaptive ? c->binWork : nullptr; T.splitBlocks = adaptive ? (uint32_t)splitCap / 4u : 0u;
  const uint32_t superTiles = ((tilesX + 7u) / 8u) * ((tilesY + 7u) / 8u);
  const uint32_t grid = T.splitBlocks + ((tilesX ? ((superTiles + 7u) / 8u) * 8u * 64u : (((numBins + 3u) / 4u + 7u) / 8u) * 8u) << T.sliceShift);   // virtual blocks, a multiple of 8
  T.totalItems = grid * 4u;      // one item per (bin, slice), in the order single-wave workgroups would have been dispatched in
  T.top0 = (const float4*)c->mesh[0].top; T.topCount0 = have0 && c->mesh[0].top ? c->mesh[0].topCount : 0u;
  T.top1 = (const float4*)c->mesh[1].top; T.topCount1 = have1 && c->mesh[1].top ? c->mesh[1].topCount : 0u;
  if (T.topCount0 > RT_TOP_SLOT0 || T.topCount1 > RT_TOP_SLOT1) { setError("launchTrace: tree tables of %u / %u nodes exceed the LDS slots", T.topCount0, T.topCount1); return -1; }
  // Which variant.  A launch of a frame gets ONE workgroup of traceWaves (12, see steerTraceWaves) waves per CU: see the kernel.
  // Below RT_WIDE_RAYS rays two such launches are in flight (capi.hip rtggx_ray_trace).  A launch with fewer than RT_TINY_RAYS rays
  // lasts as long as its longest chain of dependent steps and wants every wave slot at once: single-wave workgroups without the
  // table, one per item, as in round 1 (256x144, 9 000 rays: 0.050 ms per frame against 0.063; 1920x171, 13 000 rays: 0.063 against
  // 0.071; from 40 000 rays on the resident workgroups are as fast or faster -- a strip of the 1080p frame with 70 000-150 000 rays:
  // 0.076 / 0.087 ms for the slowest of 8 / 4 strips against 0.082 / 0.092; profiles/r02_j_resident_trace.txt section 5).
  static const int forced = getenv("RTGGX_TRACE_WAVES") ? atoi(getenv("RTGGX_TRACE_WAVES")) : 0;      // measurement: 1, 10, 12, 14, 16
  const uint32_t waves = forced ? (uint32_t)forced : (!countRays || c->lastFrameRays < RT_TINY_RAYS) ? 1u : c->traceWaves;
  const uint32_t perCu = waves == 1u ? RT_SPILL_WAVES / 256u : 1u;      // single-wave workgroups: one per item, the dispatcher deals them -- up to RT_SPILL_WAVES of them (three times the
                                                                         // wave slots the chip offers this kernel); beyond that they take a second item, a third, ... (the loop in the kernel)
  if (waves == 1u) T.topCount0 = T.topCount1 = 0u;
  T.stamps = waves == 1u ? nullptr : c->traceStamps; T.launch = c->traceStampLaunch++;      // (thousands of workgroups stamping one word would take longer than the launch)
  // as many workgroups as stay resident, a multiple of 8 so that every XCD gets its share; fewer when there is less to do
  const uint32_t wanted = (T.totalItems + waves - 1u) / waves;
  uint32_t blocks = c->numCUs * perCu < wanted ? c->numCUs * perCu : wanted;
  blocks = (blocks + 7u) & ~7u;
  if (blocks * waves > RT_SPILL_WAVES) blocks = (RT_SPILL_WAVES / waves) & ~7u;      // (every wave of the launch owns a piece of the spill area)
  const FrameParams* const dfp = c->dParams + c->slot;
#define RT_LAUNCH_TRACE(W, PER_SIMD, TOP) { if (start || stop) hipExtLaunchKernelGGL((traceKernel<W, PER_SIMD, TOP>), dim3(blocks), dim3(64 * W), 0, s, start, stop, 0, dfp, T); \
                                            else hipLaunchKernelGGL((traceKernel<W, PER_SIMD, TOP>), dim3(blocks), dim3(64 * W), 0, s, dfp, T); }
  switch (waves) {
    case 1u: RT_LAUNCH_TRACE(1, 5, 0) break;
    case 10u: RT_LAUNCH_TRACE(10, 4, 1) break;
    case 12u: RT_LAUNCH_TRACE(12, 4, 1) break;
    case 14u: RT_LAUNCH_TRACE(14, 4, 1) break;
    case 16u: RT_LAUNCH_TRACE(16, 4, 1) break;
    default: setError("launchTrace: no kernel variant with %u waves", waves); return -1;
  }
#undef RT_LAUNCH_TRACE
  const uint32_t counterMask = 15u;
  if (countRays && !c->rayCountersInFlight && (c->traceLaunches < 8u || (c->traceLaunches & counterMask) == 0u)) {     // the first frames, then every 16th: ray counters and split demand, for later launches
    RT_HIP(hipMemcpyAsync(c->hostRayCounters, c->rayCounter32, 256 * 4, hipMemcpyDeviceToHost, s));
    RT_HIP(hipMemcpyAsync(c->hostRayCounters + 256, c->splitCount, 4, hipMemcpyDeviceToHost, s));
    RT_HIP(hipMemcpyAsync(c->hostRayCounters + 258, c->traceStamps + 6, 16, hipMemcpyDeviceToHost, s)); c->traceSampleLaunch = c->traceLaunches;      // sum of the kernel's durations so far, start of this launch
    RT_HIP(hipEventRecord(c->evRayCounters, s));
    c->rayCountersInFlight = true;
    // Priming: the first launches of a context wait for their own counters, so that the decisions they feed (waves per bin,
    // capacity of the split list, the stream of the visibility pass) are settled after RT_PRIME_LAUNCHES frames instead of
    // whenever the copies happen to arrive on a host that runs three frames ahead.
    if (c->traceLaunches < RT_PRIME_LAUNCHES) RT_HIP(hipEventSynchronize(c->evRayCounters));
  }
  if (countRays) ++c->traceLaunches;
  RT_HIP(hipGetLastError());
  return 0;
}

}  // namespace rt
