// Context of librtggx: device memory, streams and per-frame state behind the C ABI of rtggx.h.
//
// HBM layout (all buffers are linear, row-major, one element per pixel, W*H elements):
//   visDepth   u64   (D24 << 32) | visibility word -- the visibility pass resolves depth order
//                    with one 64-bit atomicMin per fragment; consumers read the halves
//   normal     u32   R10G10B10A2_UNORM      roughMetal u16 R8G8_UNORM     velocity u32 R16G16_FLOAT
//   rtRefl/rtDiff u32 R11G11B10_FLOAT       tss[2], fltRfl, fltDff u64 R16G16B16A16_FLOAT
//   backbuffer u32   R8G8B8A8_UNORM
// visDepth, normal, roughMetal, velocity, rtRefl, rtDiff and the ray bins exist RT_SETS times ("input sets"): stream B
// (visibility, ray generation, traversal) fills one set while the main stream (shading, denoise, tone map) still
// reads an earlier one.
// Scene: per mesh 24-byte vertices, u32 indices, 64-byte binary BVH nodes, their 128-byte 4-wide collapse,
// 64-byte leaf triangles; environment as RGBA16F mip-major (6 faces per mip); 9 float3 SH coefficients.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string>
#include <utility>
#include <vector>
#include "rtggx_device.h"

// Input sets (G-buffer, traced images, ray bins): RT_SETS of them, used round-robin.  Stream B fills set i while the main stream
// still consumes the set of the frame before; before stream B is given the work that overwrites a set, the HOST waits
// for the event of that set's last reader, three frames back -- the frames-in-flight fence of the sample
// (RayTracedGGX.cpp: FrameCount = 3), and cheaper than a cross-queue wait on the GPU (10 us per frame on stream B's
// chain).  The frame constants live in a ring of RT_SLOTS device slots: one more than sets, because the tone map of
// frame f still reads its slot after the event of set f has completed.
// Round 2: FOUR sets (the sample's FrameCount is 3).  A frame's latency through the three stages (+ the refit of a deforming mesh) is
// 0.45-0.6 ms; with three sets in flight a thin strip or a deforming mesh ran out of frames to overlap: 1920x171 strip 0.0726 -> 0.0579 ms,
// deforming bunny 0.2432 -> 0.2347, 4K 0.758 -> 0.743, the 1080p frame unchanged; five and six give nothing more (profiles/r02_c_ab_pipeline.txt).
#ifndef RT_SETS
#define RT_SETS 4
#endif
#define RT_SLOTS (RT_SETS + 1)
// The visibility target (visDepth) exists twice more than the other members of an input set: ray generation of frame f clears the
// target of frame f + 2 on its way (raytrace.hip) -- one kernel launch and one pass over 8 bytes per pixel less per frame --, and the
// target it clears must be one nobody reads any more: frame f + 2 - RT_VIS_RING's, whose last reader the host has waited for (evSetRead).
// (Round 3 cleared the target of frame f + 1, which chained the visibility pass of frame f + 1 behind ray generation f on one stream;
// round 4 runs the visibility pass on the geometry stream beside it: capi.hip placeFrame.)
#define RT_VIS_RING (RT_SETS + 2)
#define RT_VIS_CLEAR 0x00FFFFFF00000000ull      // (D24 = 1.0) << 32 | nothing drawn
namespace rt {

// What a build tells the host: written by the build's kernels, copied to pinned memory behind its last one.
#define RT_TREELET_LEVELS 3
struct BuildResult {
  uint32_t numRounds;                        // PLOC rounds (= entries of roundBase - 1)
  uint32_t treelets[RT_TREELET_LEVELS];      // refit treelets per level (lbvh.hip "treelets")
  uint32_t topCount;                         // entries of the LDS table of the tree's top
  uint32_t depth;                            // deepest leaf (number of ancestors)
  uint32_t stack4, depth4, nodes4;           // the 4-wide collapse (lbvh.hip roots4Kernel): the most entries a traversal's stack can hold, its levels, its nodes
  uint32_t error;                            // bit 0: more than RT_MAX_ROUNDS rounds; bit 1: a treelet level beyond RT_TREELET_LEVELS would be needed
  uint32_t itemCursor, roundCursor;
  uint32_t finalEntry, finalLds;             // statistics of plocFinal: clusters handed to it, clusters when its LDS rounds began
  float cost, pad2;                          // sum of the node-box half-areas (SAH cost up to constants)
};
// Everything a build derives from ONE vertex shape that refits of later shapes keep using (lbvh.hip): the binary topology (PLOC creates
// nodes in rounds; a node's children are leaves or nodes of EARLIER rounds, and node indices are handed out round by round), the
// per-primitive and per-node boxes of the latest refit, the refit schedule (treelets of <= 1024 nodes, one workgroup each, level by
// level), the list of nodes at the tree's top.  All of it is produced on the device without a host round trip; the host learns a
// handful of counts (BuildResult) when the build has ended.
struct BvhTopo {
  uint32_t* order = nullptr;     // leaf slot -> primitive (Morton order)
  int32_t *left = nullptr, *right = nullptr, *nodeParent = nullptr, *leafParent = nullptr;
  float *nodeBox = nullptr, *triBox = nullptr;
  uint32_t* cnt[RT_TREELET_LEVELS] = {};     // per node: nodes of its subtree that treelet level l still has to place
  uint32_t* roundBase = nullptr;             // device: first node of PLOC round k
  void *dTreelets = nullptr, *dRefitItems = nullptr; uint32_t* dRefitRounds = nullptr; int32_t* treeletRoots = nullptr;
  int32_t *topList = nullptr, *topRank = nullptr;
  float4* cost4 = nullptr;       // the collapse's dynamic programme per binary node: F(n, 1..3) and the choices (lbvh.hip collapseCostTreelets)
  int4* ent4 = nullptr; uint32_t* lvl4 = nullptr;      // the 4-wide collapse: per binary node the entries it has as a 4-wide node, and its level there (~0: it is not one)
  BuildResult* dResult = nullptr; BuildResult* hResult = nullptr;      // device record, pinned copy
  BuildResult result{};                      // the host's copy, valid once the build has ended
  uint32_t numTris = 0; int32_t root = -1;
  bool refittable = false;                   // a PLOC build (the Karras radix tree of RTGGX_BVH_RADIX_TREE has no rounds)
};

struct MeshDev {
  float* verts = nullptr;        // 6 floats per vertex: the buffer of the CURRENT input set (selectSet)
  // A mesh whose vertices change per frame (rtggx_refit_as) keeps one vertex buffer per input set: the shading of frame f on the
  // main stream still reads the vertices of frame f while stream B already moves them for frame f + 1.  A static mesh has one
  // allocation, and the three pointers alias it.
  float* vertsBuf[RT_SETS] = {};
  // "Fat triangles": per primitive its three 24-byte vertices side by side (80 bytes = 5 x 16: p0 n0 p1 n1 p2 n2 + 8 pad), indexed by
  // primitive id.  Ray generation and hit shading fetch a triangle's vertices with five 16-byte loads in ONE dependent step instead of
  // three index loads followed by eighteen 4-byte loads (two steps): these kernels are latency-bound (profiles/r02_d_limiter.txt).
  // Per input set like the vertices; rebuilt from them by a refit.
  float4* fat = nullptr; float4* fatBuf[RT_SETS] = {};
  uint32_t vertsVersion[RT_SETS] = {}, version = 0, latestSet = 0;
  bool deforming = false;
  float* stage[RT_SLOTS] = {};   // pinned host staging ring for the vertices handed to rtggx_refit_as
  uint32_t stageNext = 0; int pendingStage = -1;
  float* deviceStage[RT_SLOTS] = {};   // device-side staging ring of rtggx_refit_as_device; the events that order it against the caller's stream
  uint32_t deviceStageNext = 0; int pendingDeviceStage = -1; hipEvent_t evProduced = nullptr, evStaged = nullptr;
  // what a build derives from one vertex shape and a refit keeps: `topo` (BvhTopo above); `job`: a build in progress (lbvh.hip BuildJob)
  BvhTopo topo;
  struct BuildJob* job = nullptr;
  bool wantRebuild = false;      // the refitted tree's cost has drifted: the next frame starts a rebuild
  uint32_t topoVersion = 0, topoVersionOfSet[RT_SETS] = {};      // a set whose tree arrays were emitted for an older topology is emitted in full by its next refit
  float* dCost = nullptr;        // device: sum of the node-box half-areas of the current tree (SAH cost up to constants)
  float* hCost = nullptr;        // pinned: its copy, refreshed asynchronously after every refit
  hipEvent_t evCost = nullptr; bool costInFlight = false;
  float builtCost = 0.0f, lastCost = 0.0f;
  uint32_t refits = 0, rebuilds = 0;
  uint32_t* indices = nullptr;
  uint32_t numVerts = 0, numIndices = 0, numTris = 0;
  BvhNode* nodes = nullptr;      // numTris - 1 (0 when numTris == 1): the binary LBVH as built
  Bvh4Node* nodes4 = nullptr;    // same count, sparse: the 4-wide collapse the trace kernel walks
  BvhTri* tris = nullptr;        // numTris, leaf (Morton) order
  // the three above are those of the CURRENT input set: like the vertices, the boxes and leaf triangles of a deforming mesh exist
  // once per input set (frame f + 1's refit on stream R writes its set while frame f's traversal still walks the other)
  BvhNode* nodesBuf[RT_SETS] = {}; Bvh4Node* nodes4Buf[RT_SETS] = {}; BvhTri* trisBuf[RT_SETS] = {};
  // the top of the tree once more, breadth-first, for the trace kernel's LDS (rtggx_device.h RT_TOP_*; lbvh.hip planTopKernel / emitTop):
  // topo.topList[k] = node of rank k, topo.topRank[node] = its rank or -1; the table itself per input set like the nodes
  uint32_t topCount = 0; uint32_t topCountBuf[RT_SETS] = {};
  Bvh4Node* top = nullptr; Bvh4Node* topBuf[RT_SETS] = {};
  int32_t root = -1;             // 0, or ~0 for a single-triangle mesh
  uint32_t depth = 0;            // deepest leaf (number of ancestors)
  uint32_t stack4 = 0;           // the most entries a traversal of the 4-wide tree can have on its stack (BuildResult::stack4 of the latest topology)
  float bmin[3] = {0, 0, 0}, bmax[3] = {0, 0, 0};   // vertex bounds (Morton normalisation box)
};

struct EnvDev {
  uint2* texels = nullptr;       // RGBA16F, mip-major, 6 faces per mip
  uint32_t size = 0, mips = 0;
  uint32_t mipOffset[16] = {};   // texel offset of mip m (face 0)
  uint64_t totalTexels = 0;
};

// Everything the per-frame kernels need, passed by value as one kernel argument.
struct FrameParams {
  RtggxCBGlobal g;
  RtggxRayGenConstants rg;
  RtggxCBPerObject po[2];
  RtggxCBMaterial mat;
  float invWorld[2][16];         // TLAS: world -> object, row-vector row-major
  uint32_t W, H;
  uint32_t rowBegin, rowEnd;     // strip of the frame this context renders
  uint32_t flags, pad[3];        // RT_FLAG_*
};
#define RT_FLAG_VNDF 1u          // rtggx_set_sampler: visible-normal sampling of the reflection lobe instead of the reference's NDF sampling

// Scene pointers as the trace/shade kernels see them (device-resident copy in rtggx_context::dScene).
struct Scene {
  const float* verts[2]; const uint32_t* idx[2];
  const BvhNode* nodes[2]; const BvhTri* tris[2]; int32_t root[2];
  const uint2* env; uint32_t envSize, envMips; uint32_t mipOffset[16];
  const float* sh; const float* cosSin;
};

}  // namespace rt

// The 4-wide collapse's objective (lbvh.hip): cost of a 4-wide node = AREA x half-area / the root's + TRIS x triangles / all triangles.
#ifndef RT_COLLAPSE_AREA_WEIGHT
#define RT_COLLAPSE_AREA_WEIGHT 1.0f
#define RT_COLLAPSE_TRIS_WEIGHT 0.0f
#endif
#define RT_MAX_PEERS 16      // ranks whose history images a context can map (rtggx_set_history_peers): one node has eight GPUs

struct rtggx_context {
  int device = 0;
  uint32_t W = 0, H = 0;
  uint32_t rowBegin = 0, rowEnd = 0;
  uint32_t historyApron = 18;           // rows of TemporalSSOut beyond the strip the caller delivers between frames (rtggx_set_history_apron)
  uint32_t* histReach = nullptr;        // device word: the furthest a history tap reached beyond them, in rows (temporalKernel)
  hipStream_t streamMain = nullptr, streamAS = nullptr, ownMain = nullptr;
  hipStream_t streamRefit = nullptr;               // stream R: vertex uploads and tree refits of deforming meshes (rtggx_refit_as)
  // Frame pipeline (capi.hip): 1 = three stages on three streams -- C: visibility + ray generation, B: traversal, main: shading +
  // denoiser + tone map -- so that ray generation of frame f + 1 runs beside the traversal of frame f; 0 = the round-1 arrangement
  // (ray generation and traversal on B; the visibility pass on C only where launches are small).  RTGGX_PIPELINE overrides.
  int pipeline = 1;
  bool refitIssued = false; bool traceRecorded[4] = {}; hipStream_t genStream = nullptr;   // per-frame issue state (capi.hip)
  uint32_t frameCounter = 0;                       // frames started (rtggx_render_visibility); parity selects binWork / ray counters
  hipEvent_t evGenRing[4] = {};                    // ray generation of frame f done: [f & 3] (C -> B: traversal f; C -> R: the visibility pass of frame f + 2, whose target and lists it cleared)
  uint32_t genFrame[4] = {}; hipStream_t genStreamOf[4] = {};      // the frame whose ray generation evGenRing[k] belongs to (0: none), and its stream
  hipEvent_t evTraceRing[4] = {};                  // traversal of frame f done: [f & 3] (B -> main; B -> C two frames later: binWork)
  hipStream_t ownAS = nullptr, ownVis = nullptr;   // the context's own stream B / stream C; streamAS / streamVis alias streamMain / null while
  bool asyncCompute = true;                        // rtggx_set_async_compute(0) is in force (the sample's [A] toggle: one queue, submission order)
  // Launches with few rays (thin strips, small frames) leave most of the machine idle and last as long as stream B's chain
  // of dependent kernels: there the visibility pass of frame f+1 runs on a stream of its own (C), beside the traversal
  // of frame f, instead of behind it.  (On full frames the machine is saturated and this gains nothing.)
  hipStream_t streamVis = nullptr;
  hipEvent_t evVis = nullptr;           // completes with the last kernel of the most recent visibility pass, on either stream
  hipStream_t evVisStream = nullptr;    // the stream the most recent visibility pass ran on (null: none yet)
  bool lastTraceAdaptive = false;
  bool attachEvents = true;             // RTGGX_ATTACH_EVENTS=0: record the cross-stream events with hipEventRecord instead
  hipEvent_t evAS = nullptr;      // constants uploaded (stream B -> main)
  hipEvent_t evRefit = nullptr;   // vertices of the current set uploaded and the tree refitted (stream B -> stream C)
  hipEvent_t evRT = nullptr, evSetRead[RT_SETS] = {};   // ray trace done (stream B -> main); last reader of input set i done (the HOST waits for it before stream B is given work that overwrites the set)
  bool setReadRecorded[RT_SETS] = {};
  int setReadDeferred = -1;      // the set whose event is still to ride on a later kernel of this frame (capi.hip settleSetRead)
  double fenceWaitUs = 0.0; uint32_t fenceWaits = 0;      // host time spent waiting at the frames-in-flight fence (rtggx_render_visibility; rtggx_debug_fence_wait)
  // The temporal pass and the tone map as one kernel (denoise.hip temporalToneKernel): rtggx_denoise then writes the back buffer as well and
  // the rtggx_tone_map that follows it in the same frame has nothing left to launch.  -1: where it pays -- small launches (capi.hip
  // rtggx_denoise); rtggx_debug_fuse_tone_map(ctx, 0 / 1): never / always.
  int fuseToneMap = -1; bool toneMapDone = false, denoiseIssued = false;
  // Multi-GPU strips: every rank's two history images as mapped into THIS process (rtggx_set_history_peers) -- device table
  // [2][RT_MAX_PEERS] pointers + [RT_MAX_PEERS + 1] row boundaries; a history tap beyond the exchanged apron reads the owner's image.
  uint32_t* exchangeTokens = nullptr;      // RTGGX_BUF_EXCHANGE_TOKENS
  float collapseWeights[2] = {RT_COLLAPSE_AREA_WEIGHT, RT_COLLAPSE_TRIS_WEIGHT};      // rtggx_debug_collapse_weights
  uint32_t peerWorld = 0; void* dPeerTable = nullptr; std::vector<void*> ipcMapped;      // (what rtggx_history_ipc_open mapped: unmapped by rtggx_destroy)
  uint32_t lastPlacement[2] = {0, 0};      // key and placement of the most recent rtggx_ray_trace (rtggx_debug_placement)
  int forcePlacement = -1;       // rtggx_debug_placement: -1 by the ray count; 0 / 1: the placement of a full-size / a small launch whatever the count
  bool fltRflIsFltDff = false;          // the last denoise ran without diffuse passes: FilteredOut == FilteredOut1 and only the latter was written
  bool externalStream = false;

  bool vndf = false;             // rtggx_set_sampler
  float rebuildRatio = 1.2f; uint32_t rebuildSteps = 16;      // rtggx_set_refit_policy
  rt::MeshDev mesh[2];
  rt::EnvDev env;
  float* sh = nullptr;           // 27 floats
  float* cosSinTab = nullptr;    // 512 floats: cos[256], sin[256]

  // render targets
  // Everything the visibility and ray-tracing passes (stream B) write and the denoiser (main stream) reads exists
  // RT_SETS times, so that frame N+1's visibility + ray trace overlap frame N's denoise + tone map.  The unsuffixed
  // pointers are set `setIndex`, the frame being rendered (advanced by rtggx_render_visibility).
  unsigned long long* visDepth = nullptr;
  uint32_t *normal = nullptr, *velocity = nullptr, *rtRefl = nullptr, *rtDiff = nullptr, *backbuffer = nullptr;
  uint16_t* roughMetal = nullptr;
  unsigned long long* visDepthBuf[RT_VIS_RING] = {};      // by frameCounter % RT_VIS_RING
  // which target the last ray generation cleared for the next frame's visibility pass, and over which rows (visibility.hip)
  struct VisCleared { uint32_t frame = 0, rows[2] = {0, 0}; } visClearedAt[RT_VIS_RING];      // target k has been cleared, over these rows, FOR this frame (0: not)
  uint32_t visStandaloneClears = 0;
  // Round 4: one word per 16x16 tile and target, set by the rasterisers where they draw: a tile whose word is 0 holds nothing but the clear
  // value, and ray generation neither reads nor re-clears it (three quarters of the bunny frame: 16 bytes per pixel and the first of its
  // dependent fetches).  Tiles are ray generation's, counted from the pass's first row: the words mean something only for the rows they
  // were kept under (visFlags[k].rows; another strip: everything is read and cleared, which also resets the words).
  uint32_t* visDirtyBuf[RT_VIS_RING] = {};
  struct VisFlags { uint32_t rows[2] = {0, 0}; uint32_t rasterFrame = 0; } visFlags[RT_VIS_RING];      // word 0 => tile clear, for tiles counted from rows[0]; rasterFrame: the frame whose visibility pass drew into the target last
  uint32_t* depth32 = nullptr; uint32_t* depth32Buf[RT_SETS] = {};      // the D24 word of visDepth once more, 4 bytes per pixel, for the spatial filters (written by ray generation)
  uint32_t *normalBuf[RT_SETS] = {}, *velocityBuf[RT_SETS] = {}, *rtReflBuf[RT_SETS] = {}, *rtDiffBuf[RT_SETS] = {};
  uint16_t* roughMetalBuf[RT_SETS] = {};
  uint32_t setIndex = 0;
  void *rayQueueBuf[RT_SETS] = {}, *hitQueueBuf[RT_SETS] = {};   // ray bins: written on stream B, shaded on the main stream
  uint32_t* binCountBuf[RT_SETS] = {};
  void selectSet(uint32_t i) {
    setIndex = i; visDepth = visDepthBuf[frameCounter % RT_VIS_RING]; depth32 = depth32Buf[i]; normal = normalBuf[i]; velocity = velocityBuf[i]; rtRefl = rtReflBuf[i]; rtDiff = rtDiffBuf[i]; roughMetal = roughMetalBuf[i];
    rayQueue = rayQueueBuf[i]; hitQueue = hitQueueBuf[i]; binCount = binCountBuf[i];
    splitList = splitListBuf[i]; splitCount = largeCountBase ? largeCountBase + 2 + i : nullptr;
    largeTris = largeTrisBuf[frameCounter & 1u]; largeCount = largeCountBase ? largeCountBase + (frameCounter & 1u) : nullptr;
    for (auto& m : mesh) { m.verts = m.vertsBuf[i]; m.fat = m.fatBuf[i]; m.nodes = m.nodesBuf[i]; m.nodes4 = m.nodes4Buf[i]; m.top = m.topBuf[i]; m.topCount = m.topCountBuf[i]; m.tris = m.trisBuf[i]; }
    const uint32_t par = pipeline != 0 ? (frameCounter & 1u) : 0u;
    binWork = binWorkBuf[par]; rayCounter32 = rayCounterBuf + (pipeline != 0 ? (frameCounter & 3u) : 0u) * 256u;
  }
  uint2 *tss[2] = {nullptr, nullptr}, *fltRfl = nullptr, *fltDff = nullptr;
  uint32_t frameParity = 0;

  // visibility scratch
  // LargeTri records queued by rasterSmall, merged by rasterLarge; twice, by frame parity (ray generation of frame f empties the
  // list of frame f + 1: the count it zeroes must not be the one a consumer of frame f could still read)
  void* largeTris = nullptr; void* largeTrisBuf[2] = {};
  uint32_t* largeCount = nullptr;       // the current frame's count (selectSet)
  uint32_t* largeCountBase = nullptr;   // [0], [1] entries of largeTrisBuf[parity]; [2 + set] entries of splitList[set] (zeroed by the previous frame's ray generation)
  // Bins whose traversal was expensive in the previous frame are traced by 2, 4 or 8 waves (trace.hip "adaptive split"):
  uint32_t* binWork = nullptr;          // [numBinsMax] lane-steps the trace kernel spent on the bin (read and zeroed by rayGenKernel)
  uint32_t* binWorkBuf[2] = {};         // by frame parity: ray generation of frame f reads what the traversal of frame f - 2 recorded
                                        // (frame f - 1's may still be running beside it) and the traversal of frame f records anew
  // per input set (the visibility pass of the next frame, which empties its set's list, may run beside this frame's traversal):
  // the words of the current frame's target, for the kernels that follow its visibility pass: the target's own where they describe rows [rb, re), else all ones
  const uint32_t* tileWords(uint32_t rb, uint32_t re) const {
    const VisFlags& vf = visFlags[frameCounter % RT_VIS_RING];
    return useTileWords && !traversalBound && vf.rasterFrame == frameCounter && vf.rows[0] == rb && vf.rows[1] == re ? visDirtyBuf[frameCounter % RT_VIS_RING] : visDirtyOnes;
  }
  bool useTileWords = true;      // rtggx_debug_tile_words
  // Where the TRAVERSAL is the frame's period (two rays per pixel into a large mesh: it runs 96 % of the time) nobody asks the words:
  // workgroups over empty tiles that leave at once make the other stages' kernels run denser beside the traversal and stretch it -- dragon,
  // metallic 0.25 / 0.5: 0.354 ms without the words, 0.360 with them everywhere but in the filters, 0.378 with them in the filters as well
  // (profiles/r04_j_tile_words.txt).  Decided once per frame (launchRayTrace) from the share of the period the traversal's own time stamps
  // measure (trace.hip steerTraceWaves), with hysteresis; the words themselves are kept either way.
  bool traversalBound = false;
  const uint32_t* traceTileWords = nullptr;      // launchRayTrace -> launchTrace: tileWords() of the frame's G-buffer rows
  uint32_t* visDirtyOnes = nullptr;      // as many words as a visDirtyBuf, all ones: "every tile may hold something" (raytrace.hip GenArgs)
  uint32_t* splitListBuf[RT_SETS] = {}; // [RT_SPLIT_CAP] (shift << 28) | (slice << 24) | bin, one entry per wave of a listed bin
  uint32_t* splitList = nullptr; uint32_t* splitCount = nullptr;     // the current set's (selectSet)
  uint32_t splitDemand = 0;             // entries the most recent frame whose count has arrived wanted (hostRayCounters[256])
  uint32_t splitCapForced = 0xFFFFFFFFu;   // rtggx_debug_trace_split: fixed capacity instead of the demand-driven one
  uint32_t splitWork = 0, splitMaxShift = 0;   // set at creation (RT_SPLIT_WORK, or RTGGX_SPLIT_WORK / RTGGX_SPLIT_MAX_SHIFT)
  uint32_t largeCapacity = 0;

  // ray bins of the trace pass (rt_queue.h): numBinsMax bins of 128 64-byte ray records + 16-byte hit records
  void* rayQueue = nullptr;
  void* hitQueue = nullptr;
  uint32_t* binCount = nullptr;         // rays in each bin
  uint32_t numBinsMax = 0, binSlots = 64;      // bins per set; ray slots per bin (rt_queue.h RT_BIN_MIN / RT_BIN)
  void *testRayRange = nullptr, *traceRayRange = nullptr;      // rtggx_trace_rays: the rays' own (TMin, TMax); set only around that entry point's launch
  int32_t* stackOverflow = nullptr;     // traversal-stack spill area (entries beyond the LDS stack), sized from
  uint32_t spillEntries = 0;            // the depth of the built trees: [spillEntries][numBinsMax * 128] words
  void* dummyRecord = nullptr;          // 128 zero bytes: record base for meshes without nodes / absent meshes
  uint32_t* dEnvMipOffset = nullptr;    // device copy of env.mipOffset
  bool lastTraceSmall = false; uint32_t traceSpillHalf = 0;
  hipStream_t shadeStream = nullptr;     // the stream the most recent hit shading ran on (capi.hip rtggx_ray_trace: the main stream, or the traversal's for small launches)
  // RayTracingOut1 keeps what it held where no diffuse ray is traced: with several input sets, carried over from the previous set -- by ray
  // generation when the previous frame's shading kernel wrote nothing into that set (genCarriesDiff), else by the shading kernel (raytrace.hip)
  bool genCarriesDiff = false, shadeWroteDiff = false, lastFrameDiffuse = false;
  uint32_t numCUs = 256;
  // the trace kernel's workgroup size (full-size launches) and the time stamps it takes of itself (trace.hip): stamps = 3 x (start, end) + (sum of durations, latest start)
  uint32_t traceWaves = 12, traceWavesForced = 0; float traceShare = 0.0f; unsigned long long* traceStamps = nullptr; uint32_t traceStampLaunch = 0;
  unsigned long long traceStampSum = 0, traceStampStart = 0; uint32_t traceStampAt = 0, traceSampleLaunch = 0;      // the previous sample; the launch the sample in flight was taken at
  uint32_t traceTrial = 0, traceCooldown = 0, traceWavesSince = 0; float traceTrialBase = 0.0f;      // a trial of two more waves: samples seen, the period to beat

  // counters
  uint32_t* hostRayCounters = nullptr;  // pinned copy of rayCounter32[0..255], refreshed asynchronously after every trace launch
  hipEvent_t evRayCounters = nullptr; bool rayCountersInFlight = false; uint32_t traceLaunches = 0;
  uint32_t lastFrameRays = 0xFFFFFFFFu; // rays of the most recent frame whose counters have arrived (unknown: assume a full machine)
  uint32_t* rayCounter32 = nullptr;     // 256 per-frame partial counts written by the trace kernel (the current frame's half of ...)
  uint32_t* rayCounterBuf = nullptr;    // ... [4][256] by frame number & 3, then 768 words of RT_TRACE_STATS counters.  (Four, not two like binWork:
                                        // ray generation of frame f zeroes its quarter, and the asynchronous copy of frame f - 2's counters to the
                                        // host, queued behind that frame's traversal, may not have run yet; frame f - 4's has.)
  uint32_t* lastRayCounter32 = nullptr; // the half of the most recent rtggx_ray_trace (rtggx_ray_count)
  unsigned long long* rayCounter = nullptr;   // [0..255] last frame, [256..511] running total

  // per-frame constants: ring of RayTracer::FrameCount slots (host side; kernels take them by value)
  rt::FrameParams slots[RT_SLOTS];
  uint32_t slot = 0;
  rt::FrameParams* dParams = nullptr;   // device ring, 3 slots; kernels read their constants from here
  rt::Scene* dScene = nullptr;          // device copy of the scene pointers
  bool sceneDirty = true, slotUploaded = false;
  RtggxCBMaterial material;
  float invWorld[2][16];
  bool haveConstants = false, asBuilt = false, shDone = false;

  // timing
  bool timing = false;        // all per-pass events (rtggx_get_timings)
  bool kernelRing = false;    // only the ray-trace kernel, one event pair per sampled frame in a ring (rtggx_kernel_times)
  uint32_t ringStride = 1, ringTick = 0;   // every ringStride-th frame is sampled
  std::vector<hipEvent_t> kevBegin, kevEnd;
  uint32_t kevCount = 0;
  hipEvent_t tev[16];
  RtggxTimings lastTimings{};
  bool timingsPending = false;
};

namespace rt {
// Rows each pass must cover so that the strip [rowBegin,rowEnd) of the final image is exact
// (SURVEY.md 8e): the tone map reads TSS at +-1 row, the temporal pass FilteredOut1 at +-1, the
// vertical filters the horizontal scratch at +-16.  Pure per-pixel passes recompute the apron
// instead of exchanging it.
enum RowPass { ROWS_GBUFFER /* visibility, ray trace, H filters: +-18 */, ROWS_VFILTER /* +-2 */, ROWS_TEMPORAL /* +-1 */, ROWS_FINAL };
inline void passRows(const FrameParams& fp, RowPass pass, uint32_t& b, uint32_t& e) {
  const uint32_t apron = pass == ROWS_GBUFFER ? 18u : pass == ROWS_VFILTER ? 2u : pass == ROWS_TEMPORAL ? 1u : 0u;
  b = fp.rowBegin > apron ? fp.rowBegin - apron : 0u;
  e = fp.rowEnd + apron < fp.H ? fp.rowEnd + apron : fp.H;
  if (fp.rowEnd <= fp.rowBegin) { b = e = 0; }
}
void setError(const char* fmt, ...);
#define RT_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { rt::setError("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); return -2; } } while (0)

// kernels / launchers implemented in the .hip files
int uploadParams(rtggx_context* c, uint32_t slot, hipStream_t s);
int uploadScene(rtggx_context* c, hipStream_t s);
int launchVisibility(rtggx_context* c, const FrameParams& fp, hipStream_t s, hipEvent_t done = nullptr);
// Acceleration-structure builds (lbvh.hip).  A build is a fixed sequence of kernel launches on one stream, no host round trip in it:
//   buildLbvh          all of it at once, then ONE wait (rtggx_build_as; the sample: BuildAccelerationStructures + one WaitForGpu)
//   startRebuild       the same sequence for a mesh that deforms, from the vertices of input set `set`, into a topology of its own ...
//   continueRebuild    ... issued a few launches per frame behind the frame's refit; when the last one has ended (an event the host
//                      polls) the new topology replaces the old one between two frames.  Nothing waits.
int buildLbvh(rtggx_context* c, uint32_t slot, hipStream_t s);
int startRebuild(rtggx_context* c, uint32_t slot, uint32_t set);      // 1: started, 0: not (one is in progress, or the mesh cannot be refitted), < 0: error
int continueRebuild(rtggx_context* c, uint32_t slot, hipStream_t s, uint32_t maxSteps, bool* swapped);      // maxSteps 0: only ask whether a build whose launches are all out has ended (then the swap); > 0: only issue launches
int prepareRebuild(rtggx_context* c, uint32_t slot);      // the second topology and the build's scratch memory, once, when a mesh begins to deform
void abandonRebuild(rtggx_context* c, uint32_t slot);      // (synchronises; before the mesh's buffers are freed)
int refitLbvh(rtggx_context* c, uint32_t slot, uint32_t set, hipStream_t s);      // boxes of the existing tree from the vertices of input set `set`, into that set's BVH arrays: no host round trip
void freeBuildProducts(MeshDev& m);
int buildFatTris(rtggx_context* c, uint32_t slot, uint32_t set, hipStream_t s);      // mesh.fatBuf[set] from mesh.vertsBuf[set] and the indices
int launchRayTrace(rtggx_context* c, const FrameParams& fp, hipStream_t sGen, hipStream_t sTrace, hipEvent_t done = nullptr);   // ray generation on sGen, traversal on sTrace (joined by evGen when they differ)
// `done` (may be null) on the launch functions below: an event that completes with the pass's last kernel.  It rides on that
// kernel's own completion signal (hipExtLaunchKernelGGL) instead of a marker packet behind it: a marker costs its queue
// 5-7 us, and the frame's two chains had four of them (rocprofv3 kernel trace, profiles/).
int launchShade(rtggx_context* c, const FrameParams& fp, hipStream_t s, hipEvent_t done = nullptr);      // hit / miss shading of the traced bins
int launchTraceRays(rtggx_context* c, const FrameParams& fp, const float* dRays, uint32_t n, float* dOut, hipStream_t s);
int launchDenoise(rtggx_context* c, const FrameParams& fp, int useLds, hipStream_t s, hipEvent_t done = nullptr, bool fuseToneMap = false);      // fuseToneMap: the last kernel also writes the back buffer
int launchToneMap(rtggx_context* c, const FrameParams& fp, hipStream_t s, hipEvent_t done = nullptr);
int decodeEnv(rtggx_context* c, int format, uint32_t size, uint32_t mips, const void* hostData, size_t bytes, hipStream_t s);
int projectSH(rtggx_context* c, hipStream_t s);
int unpackVisDepth(rtggx_context* c, uint32_t* dVis, uint32_t* dDepth, hipStream_t s);
int packVisDepth(rtggx_context* c, const uint32_t* dVis, const uint32_t* dDepth, hipStream_t s);
}  // namespace rt
