/*
 * rtggx.h -- C ABI of librtggx, the MI355X (gfx950) implementation of the RayTracedGGX hot path.
 *
 * This is the drop-in boundary: each entry point replaces one method of the
 * reference's pass objects, minus the D3D12 handle parameters.  The reference-side binding a
 * maintainer would add is shown in INTEGRATION.md.  No C++ or torch types cross this boundary;
 * every function returns 0 on success or a negative code (text via rtggx_last_error()).
 * A context is used from one host thread and owns all device memory; host arrays passed to
 * rtggx_set_* are copied before the call returns.
 *
 *   reference interface (RayTracedGGX/...)                          entry point
 *   -------------------------------------------------------------   ---------------------------
 *   RayTracer::Init            Content/RayTracer.h:24-29, .cpp:66    rtggx_create + rtggx_set_mesh + rtggx_set_env
 *   Denoiser::Init             Content/Denoiser.h:15-17, .cpp:21     rtggx_create (render targets of both)
 *   createVB/createIB/createGroundMesh   RayTracer.cpp:393-511       rtggx_set_mesh
 *   DDS::Loader::CreateTextureFromFile   RayTracer.cpp:143-150       rtggx_set_env
 *   buildAccelerationStructures + BuildAccelerationStructures        rtggx_build_as
 *                              RayTracer.cpp:676-716, 158-233
 *   RayTracer::SetMetallic     RayTracer.cpp:244-248                 rtggx_set_metallic
 *   CBMaterial upload          RayTracer.cpp:129-140                 rtggx_set_material
 *   RayTracer::UpdateFrame     RayTracer.cpp:250-305 (constants)     rtggx_update_frame
 *   RayTracer::UpdateAccelerationStructure   RayTracer.cpp:326-341   rtggx_update_as
 *   RayTracer::TransformSH     RayTracer.cpp:307-310                 rtggx_transform_sh
 *   RayTracer::RenderVisibility RayTracer.cpp:343-365, 751-791       rtggx_render_visibility
 *   RayTracer::RayTrace        RayTracer.cpp:367-376, 793-810        rtggx_ray_trace
 *   Denoiser::Denoise          Denoiser.cpp:66-75                    rtggx_denoise
 *   Denoiser::ToneMap          Denoiser.cpp:77-103                   rtggx_tone_map
 *   GetRayTracingOutputs/GetGBuffers/GetDepth  RayTracer.cpp:378-391 rtggx_readback / rtggx_buffer_ptr
 *   WaitForGpu                 RayTracedGGX.cpp:672-682              rtggx_sync
 */
#ifndef RTGGX_H
#define RTGGX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rtggx_context rtggx_context;

enum { RTGGX_GROUND = 0, RTGGX_MODEL_OBJ = 1, RTGGX_NUM_MESH = 2 };   /* RayTracer::MeshIndex, RayTracer.h:13-19 */

/* Constant buffers of the reference, byte for byte.  4x4 matrices are
 * stored as the reference uploads them (XMStoreFloat4x4 of the transpose): logical row-vector
 * matrix M[i][j] = f[j*4+i].  3x4 blocks are XMStoreFloat3x4 images. */
typedef struct RtggxCBGlobal {            /* RayTracer.cpp:27-35  <->  RayTracing.hlsl:46-53 */
  float    WorldViewProjs[2][16];
  float    WorldViewProjsPrev[2][16];
  float    Worlds[2][12];
  float    WorldITs0[12];
  float    WorldIT1[11];
  uint32_t FrameIndex;
} RtggxCBGlobal;
typedef struct RtggxRayGenConstants {     /* RayTracer.cpp:20-25  <->  RayTracing.hlsl:55-60 */
  float ProjToWorld[16];
  float EyePt[4];
  float ProjBias[2];
  float pad[2];
} RtggxRayGenConstants;
typedef struct RtggxCBPerObject {         /* RayTracer.cpp:37-41  <->  VSVisibility.hlsl:17-21 */
  float WorldViewProj[16];
  float ProjBias[2];
  float pad[2];
} RtggxCBPerObject;
typedef struct RtggxCBMaterial {          /* RayTracer.cpp:43-47  <->  Material.hlsli:10-14 */
  float BaseColors[2][4];
  float RoughMetals[2][4];
} RtggxCBMaterial;
typedef struct RtggxFrameConstants {      /* 768 bytes */
  RtggxCBGlobal        global;
  RtggxRayGenConstants rayGen;
  RtggxCBPerObject     perObject[2];
  RtggxCBMaterial      material;          /* ignored by rtggx_update_frame: CBMaterial is persistent, see rtggx_set_material */
} RtggxFrameConstants;

/* Environment texel formats accepted by rtggx_set_env (DXGI numbering). */
enum { RTGGX_FORMAT_RGBA32F = 2, RTGGX_FORMAT_RGBA16F = 10, RTGGX_FORMAT_BC6H_UF16 = 95, RTGGX_FORMAT_BC6H_SF16 = 96 };

/* Buffers readable with rtggx_readback (one element per pixel unless noted). */
enum {
  RTGGX_BUF_VISIBILITY = 0,  /* uint32  ((instance<<24)|primitive)+1, 0 = empty   PSVisibility.hlsl:23 */
  RTGGX_BUF_DEPTH = 1,       /* uint32  D24 value in the low 24 bits */
  RTGGX_BUF_NORMAL = 2,      /* uint32  R10G10B10A2_UNORM */
  RTGGX_BUF_ROUGH_METAL = 3, /* uint16  R8G8_UNORM */
  RTGGX_BUF_VELOCITY = 4,    /* uint32  R16G16_FLOAT */
  RTGGX_BUF_RT_REFL = 5,     /* uint32  R11G11B10_FLOAT, RayTracingOut0 */
  RTGGX_BUF_RT_DIFF = 6,     /* uint32  R11G11B10_FLOAT, RayTracingOut1 */
  RTGGX_BUF_TSS0 = 7,        /* uint64  R16G16B16A16_FLOAT, TemporalSSOut0 */
  RTGGX_BUF_TSS1 = 8,        /* uint64  TemporalSSOut1 */
  RTGGX_BUF_FLT_RFL = 9,     /* uint64  FilteredOut (with no diffuse pass to read it -- both instances fully metallic -- it equals FilteredOut1
                                bit for bit and only that one is written: readback and buffer_ptr then return FilteredOut1) */
  RTGGX_BUF_FLT_DFF = 10,    /* uint64  FilteredOut1 */
  RTGGX_BUF_BACKBUFFER = 11, /* uint32  R8G8B8A8_UNORM */
  RTGGX_BUF_SH_COEFFS = 12,  /* 27 floats: 9 x float3 */
  RTGGX_BUF_BVH_NODES0 = 13, /* 64-byte nodes of mesh 0 (see DESIGN.md "BVH layout") */
  RTGGX_BUF_BVH_TRIS0 = 14,  /* 64-byte leaf triangles of mesh 0: v0,v1,v2 (9 floats), 3 pad, primitive id (word 12), 3 pad */
  RTGGX_BUF_BVH_NODES1 = 15,
  RTGGX_BUF_BVH_TRIS1 = 16,
  RTGGX_BUF_TLAS = 17,       /* 2 x 16 floats: world->object matrices (row-vector, row-major) */
  RTGGX_BUF_ENV = 18,        /* decoded RGBA16F environment, mip-major, 6 faces per mip */
  RTGGX_BUF_BVH4_NODES0 = 19, /* 128-byte 4-wide nodes of mesh 0, indexed like the binary nodes (the slots of binary nodes folded into another: zero): */
  RTGGX_BUF_BVH4_NODES1 = 20, /*   minx[4] miny[4] minz[4] maxx[4] maxy[4] maxz[4] ref[4] pad[4]; ref: >=0 node, <0 ~leaf slot, 0x7FFFFFFF none */
  RTGGX_BUF_BIN_WORK = 21,    /* uint32 per ray bin (8x8-pixel sub-tile; bin = 4 * (tileY * tilesX + tileX) + 2 * subY + subX over 16x16 tiles):
                                 lane-steps the last traversal spent on the bin's rays; zero unless that launch recorded them (full-size frames) */
  RTGGX_BUF_BVH4_TOP0 = 22,   /* the first (up to 16 / 96) 4-wide nodes of mesh 0 / 1 in breadth-first order, same 128-byte records; a reference to */
  RTGGX_BUF_BVH4_TOP1 = 23,   /*   a node that is in the table itself reads 0x40000000 | position (the copy the trace kernel keeps in LDS) */
  RTGGX_BUF_EXCHANGE_TOKENS = 24, /* 2 x RTGGX_MAX_PEERS uint32: words a multi-GPU host may send from ([rank]) and receive into ([RTGGX_MAX_PEERS + peer]) --
                                     the 4-byte messages that order two ranks without a neighbour's history rows between them (rtggx_set_history_peers) */
  RTGGX_BUF_COUNT = 25
};

/* Per-pass GPU timings of the last completed frame, in milliseconds (hipEvent based). */
typedef struct RtggxTimings {
  float update_as, visibility, ray_trace, spatial_refl_h, spatial_refl_v, spatial_diff_h, spatial_diff_v,
        temporal, tone_map, frame,
        ray_trace_kernel;   /* the fused raygen/trace/shade kernel alone (events right around its launch) */
} RtggxTimings;

const char* rtggx_last_error(void);

/* Creates a context on HIP device `device` with all render targets of RayTracer::Init and
 * Denoiser::Init for a width x height viewport.  The ground mesh of createGroundMesh and the
 * default materials are installed. */
int  rtggx_create(rtggx_context** out, uint32_t width, uint32_t height, int device);
void rtggx_destroy(rtggx_context* ctx);

/* Restrict rendering to the row strip [row_begin, row_end) of the full frame (multi-GPU screen
 * tiling); buffers stay full-size, rows outside the strip (plus the apron the
 * filters need) are not touched.  Default: the whole frame. */
int  rtggx_set_strip(rtggx_context* ctx, uint32_t row_begin, uint32_t row_end);

/* Strips only.  The temporal pass reprojects last frame's TemporalSSOut; rows next to a strip edge come from the neighbouring
 * rank, which the caller delivers between frames (`rows` beyond each edge; default 18 = 16 px of vertical motion per frame +
 * the bilinear tap + the pass's own 1-row apron).  rtggx_history_overreach returns the largest number of rows by which a
 * history tap read BEYOND the delivered rows since the last reset (0: every frame equals the single-GPU frame); synchronises. */
int  rtggx_set_history_apron(rtggx_context* ctx, uint32_t rows);
int  rtggx_history_overreach(rtggx_context* ctx, uint32_t* rows, int reset);

/* Make an externally owned hipStream_t the context's MAIN stream: shading, denoise and tone map run on it, and
 * every result the caller may read (traced images, filtered images, back buffer) is produced in its order.  The
 * visibility pass, ray generation and traversal keep running ahead on the context's internal stream B, joined
 * to the main stream by events.  NULL (also the handle of the null stream) restores the context's own stream. */
int  rtggx_set_stream(rtggx_context* ctx, void* hip_stream);
/* The context's main stream (its own, or the one handed in): what a host enqueues there -- the per-frame RCCL exchange of the
 * multi-GPU host, host/Strips.cpp -- is ordered behind the frame's tone map and before the next frame's temporal pass. */
int  rtggx_get_stream(rtggx_context* ctx, void** hip_stream);
/* Multi-GPU strips: history taps beyond the exchanged apron read the OWNER's image (round 4).  The reference samples its one history
 * texture anywhere (CSTemporalSS.hlsl:259-265); a rank holds last frame's TemporalSSOut for its own rows and `apron` rows either side.
 * With the other ranks' two history images mapped into this process the temporal pass reads a tap beyond those rows from the image of
 * the rank whose strip holds the row: N strips equal the single-GPU frame at any velocity (rtggx_history_overreach keeps counting such
 * taps; they are harmless then).
 *   bounds      world + 1 ascending rows: rank r owns [bounds[r], bounds[r + 1]) -- every rank allocates full-size targets, so a row sits
 *               at the same offset in every rank's image
 *   tss0, tss1  world device pointers each, valid in THIS process: rank r's TemporalSSOut[0] / [1] (rtggx_buffer_ptr of a context in
 *               the same process, or rtggx_history_ipc_open of another process's export); this rank's own entries may be null
 *   world = 0   forget the peers
 * Ordering is the caller's: rank A's temporal pass of frame f + 1 may read rank B's image once B's temporal pass of frame f has ended,
 * and B's horizontal filter of frame f + 2 -- which reuses that image as its scratch -- must wait for A's temporal pass of frame f + 1.
 * A per-frame exchange on the main streams (rtggx_get_stream) with a message in EACH direction between every two ranks orders both:
 * the neighbours' history rows do between neighbours, host/Strips.cpp and strips.py add 4-byte tokens between the other pairs. */
#define RTGGX_MAX_PEERS 16
#define RTGGX_IPC_HANDLE_BYTES 64
int  rtggx_set_history_peers(rtggx_context* ctx, uint32_t world, const uint32_t* bounds, void* const* tss0, void* const* tss1);
/* One process per GPU: this context's two history images as inter-process handles (2 x RTGGX_IPC_HANDLE_BYTES: hipIpcMemHandle_t of
 * TemporalSSOut[0], [1]; `bytes` = the room at `handles`) ... */
int  rtggx_history_ipc_export(rtggx_context* ctx, void* handles, size_t bytes);
/* ... and another process's handles opened in this one: two device pointers for rtggx_set_history_peers (unmapped by rtggx_destroy).
 * (HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment of both processes on hosts whose driver only supports dmabuf IPC.) */
int  rtggx_history_ipc_open(rtggx_context* ctx, const void* handles, size_t bytes, void** tss0, void** tss1);

/* The sample's asynchronous-compute toggle (m_asyncCompute, key [A]: RayTracedGGX.cpp:304-353 issues the frame over two
 * queues, :513-556 as one command list).  enable = 0: every pass is issued to the main stream in submission order (no
 * stream B / C, no overlap between frames); 1 (default): the multi-stream frame.  Results are bit-identical; synchronises. */
int  rtggx_set_async_compute(rtggx_context* ctx, int enable);

/* Vertex = {float3 Pos; float3 Nrm} (24 bytes), 32-bit indices, triangle list. */
int  rtggx_set_mesh(rtggx_context* ctx, uint32_t slot, const float* verts, uint32_t num_verts,
                    const uint32_t* indices, uint32_t num_indices);
/* Cube map: `size` x `size` faces, `mips` levels, `data` laid out as in a DDS file (face-major,
 * full mip chain per face, faces +X -X +Y -Y +Z -Z).  BC6H blocks are decoded on the device. */
int  rtggx_set_env(rtggx_context* ctx, int format, uint32_t size, uint32_t mips, const void* data, size_t bytes);
int  rtggx_set_material(rtggx_context* ctx, uint32_t mesh, const float base_color[4], float roughness, float metallic);
int  rtggx_set_metallic(rtggx_context* ctx, uint32_t mesh, float metallic);
/* Sampler of the reflection lobe.  0 (default): the reference's -- the GGX normal distribution itself, computeLocalDirectionGGX /
 * computeReflection (RayTracing.hlsl:92-101, 129-147, 424-484), weight NoL F Vis 4 VoH / NoH -- the parity path.  1: the distribution
 * of VISIBLE normals (Heitz 2018), weight F G1(L): no sample is wasted below the horizon of the view direction, less variance at
 * grazing angles for the same one sample per pixel.  Takes effect with the next rtggx_update_frame. */
int  rtggx_set_sampler(rtggx_context* ctx, int vndf);

/* Build of both bottom-level structures (RayTracer::buildAccelerationStructures / BuildAccelerationStructures, RayTracer.cpp:676-716,
 * 158-233; the sample records the builds on the GPU timeline and waits once, RayTracedGGX.cpp:236): every step of the build --
 * Morton codes, sort, PLOC clustering, the refit schedule, the node arrays -- is a kernel launch on the context's build stream
 * with no host round trip between them; the call waits once, at the end. */
int  rtggx_build_as(rtggx_context* ctx);

/* Deforming mesh: `num_verts` new vertices (same layout, same count, same indices as the last rtggx_set_mesh) for mesh `slot`.
 * Replaces, for shape changes, what RayTracer::UpdateAccelerationStructure (RayTracer.cpp:326-341) does for the rigid instance
 * motion of the sample: the acceleration structure follows the new positions without a synchronisation.
 * The vertices are copied (staged) before the call returns; the upload, the new leaf triangles and the bottom-up box refit of
 * the existing tree run on the context's refit stream at the start of the next frame (rtggx_render_visibility), overlapping the
 * previous frame's shading and denoising.  A refit keeps the topology of the last build; when the refitted tree's cost exceeds
 * `rebuild_ratio` x that of the last build (rtggx_set_refit_policy, default 1.2), the mesh is REBUILT from its newest shape beside
 * the frames: `steps_per_frame` kernel launches of the build per frame (default 16; ~75 for the bunny) behind that frame's refit,
 * the new topology taking over between two frames when the last one has ended.  Neither call waits for the GPU (the first
 * rtggx_refit_as of a mesh allocates its per-set buffers).  While a mesh deforms on a full-size frame the context keeps three frames in
 * flight instead of four (rtggx_render_visibility waits for the end of frame f - 3): one frame time instead of two, DESIGN.md section 9.
 * rtggx_refit_stats: cost of the current tree relative to the last
 * build, refits and rebuilds so far; synchronises (a rebuild in progress stays in progress). */
int  rtggx_refit_as(rtggx_context* ctx, uint32_t slot, const float* verts, uint32_t num_verts);
/* The same for a mesh animated ON the GPU (round 4): `device_verts` is a device pointer, `hip_stream` the stream that produces it (NULL: the
 * null stream).  Like a hipMemcpyAsync on that stream: the copy out of `device_verts` is ordered behind everything the stream holds at the
 * time of the call, and what the caller enqueues there afterwards (the next animation step) behind the copy; it runs on the context's
 * geometry stream into a device-side staging ring -- no host copy, no wait. */
int  rtggx_refit_as_device(rtggx_context* ctx, uint32_t slot, const float* device_verts, uint32_t num_verts, void* hip_stream);
int  rtggx_set_refit_policy(rtggx_context* ctx, float rebuild_ratio, uint32_t steps_per_frame);
int  rtggx_refit_stats(rtggx_context* ctx, uint32_t slot, float* cost_ratio, uint32_t* refits, uint32_t* rebuilds);

/* Per-frame constants; copied into the next slot of a ring of (input sets + 1) = 5. */
int  rtggx_update_frame(rtggx_context* ctx, const RtggxFrameConstants* constants);
/* Refreshes the TLAS (the two world->object matrices) from the constants of the current slot.  May be called before or
 * after rtggx_render_visibility of the same frame (the sample overlaps the two on different queues); rtggx_ray_trace sends
 * the refreshed constants to the device again when the visibility pass had already carried the slot there. */
int  rtggx_update_as(rtggx_context* ctx);
int  rtggx_transform_sh(rtggx_context* ctx);
/* Starts a frame: advances to the next of 4 input sets (G-buffer, traced images, ray bins, and -- for a deforming mesh -- vertices and
 * tree; the sample's RayTracer::FrameCount is 3).  All pass functions only enqueue work; this one is the frames-in-flight fence of the
 * sample (RayTracedGGX.cpp:672-701): it blocks the calling thread while the frame that last used that set, four frames back, is still
 * being read on the GPU.  A pending rtggx_refit_as is issued here. */
int  rtggx_render_visibility(rtggx_context* ctx);
int  rtggx_ray_trace(rtggx_context* ctx);
int  rtggx_denoise(rtggx_context* ctx, int use_shared_mem);
int  rtggx_tone_map(rtggx_context* ctx);

/* Round 4: rtggx_denoise's temporal pass can tone-map its result as well (one kernel instead of two: Denoiser::Denoise and ::ToneMap
 * follow each other in every frame of the sample, RayTracedGGX.cpp:341-350); the rtggx_tone_map that follows it in the same frame then
 * finds its work done.  The library does so where it pays: on small launches (thin strips, small frames), not on full-size frames
 * (measured: profiles/r04_c_pipeline_ab.txt).  A tone map without a preceding rtggx_denoise in the frame, or after an rtggx_upload,
 * always runs as a kernel of its own.  Diagnostic: mode 0 = always two kernels, 1 = always fused, -1 = the library's choice again;
 * the back buffer and TemporalSSOut are bit-identical either way. */
int  rtggx_debug_fuse_tone_map(rtggx_context* ctx, int mode);
/* Diagnostic: which streams a frame's kernels go to is decided from five facts (capi.hip placeFrame: small launch, strip, deforming
 * mesh, diffuse rays, caller-owned main stream).  force_small = 0 / 1 pins the first of them whatever the ray count says (-1: by the
 * count again).  key / where (may be null): the most recent rtggx_ray_trace's key (bit 0 small, 1 strip, 2 deforming, 3 diffuse,
 * 4 caller-owned stream) and placement (bits 0-3 / 4-7 / 8-11 / 16-19: stream of ray generation / traversal / hit shading / the visibility pass --
 * 0 main, 1 B, 2 C, 3 R --, bits 12-15 frames in flight).  Results do not depend on any of it. */
int  rtggx_debug_placement(rtggx_context* ctx, int force_small, uint32_t* key, uint32_t* where);
/* Diagnostic (round 4): the visibility pass keeps one word per 16x16 tile of its target -- "something was drawn here" -- and the kernels
 * behind it (ray generation, traversal, hit shading, the tiled spatial filters) leave a tile whose word is 0 after a scalar load instead of
 * fetching pixels to find that out; ray generation neither reads nor re-clears such a tile of the target (RayTracer.cpp:751-791 clears and
 * reads the whole target every frame).  enable = 0: every tile is treated as drawn, as in rounds 1-3.  Same images either way. */
int  rtggx_debug_tile_words(rtggx_context* ctx, int enable);
/* Diagnostic: the two weights of the 4-wide collapse's objective (lbvh.hip "the 4-wide collapse"): a 4-wide node costs
 * area_weight x (its half-area / the root's) + tris_weight x (its triangles / all triangles) -- the chance that a random ray enters it,
 * and the chance that a ray STARTING on the mesh's surface (every ray of this path does) starts inside it.  set (may be null): weights for
 * builds from now on; get (may be null): the current ones.  Results do not depend on them. */
int  rtggx_debug_collapse_weights(rtggx_context* ctx, const float* set, float* get);
/* Diagnostic: the host time (us) rtggx_render_visibility has spent WAITING at the frames-in-flight fence -- for the last reader of the
 * input set it is about to overwrite, four frames back (RayTracedGGX.cpp:672-701) -- and how many frames had to wait, since the last reset.
 * A frame loop that is bound by the GPU waits there every frame; one that is bound by its own submission never does. */
int  rtggx_debug_fence_wait(rtggx_context* ctx, double* us_total, uint32_t* waits, int reset);

int  rtggx_sync(rtggx_context* ctx);
/* Number of non-degenerate rays (TMax > TMin) traced by the last rtggx_ray_trace; synchronises. */
int  rtggx_ray_count(rtggx_context* ctx, uint64_t* rays);
/* Rays traced since the last reset (accumulated on the device, no per-frame synchronisation); synchronises. */
int  rtggx_ray_total(rtggx_context* ctx, uint64_t* rays, int reset);
/* Diagnostic counters (non-zero only in builds with -DRT_TRACE_STATS): [0] lane node steps, [1] lane leaf steps,
 * [2] wave iterations, [3] refills of the trace kernel since the last reset. */
int  rtggx_debug_counters(rtggx_context* ctx, uint32_t* out, uint32_t n, int reset);
/* Tuning / test hook of the trace kernel's adaptive split (bins that were expensive in the previous frame are traced by
 * 2, 4 or 8 waves): work_per_wave = lane-steps of traversal per wave above which a bin is split further (0: never),
 * max_shift = log2 of the most waves per bin (0..3), capacity = room in the split list, in waves (-1: sized from the
 * demand of earlier frames, the default).  *last_demand (may be NULL) receives the number of list entries the most recent
 * frame asked for; synchronises.  Results do not depend on any of this: hits merge with a 64-bit atomic min. */
int  rtggx_debug_trace_split(rtggx_context* ctx, uint32_t work_per_wave, uint32_t max_shift, int capacity, uint32_t* last_demand);
/* The traversal kernel keeps one workgroup of 12 to 16 waves per CU resident on full-size frames; the size follows the share of
   the frame period the traversal takes (DESIGN.md "The trace kernel").  Reports the size and the share last sampled (0 before the
   first sample).  force_waves: 0 leaves the choice to the library, 10/12/14/16 pins it (measurement). */
int  rtggx_debug_trace_residency(rtggx_context* ctx, uint32_t force_waves, uint32_t* waves, float* share);
int  rtggx_get_timings(rtggx_context* ctx, RtggxTimings* out);
/* mode 0 off, 1 every pass (rtggx_get_timings), 2 only the ray-trace kernel: one HIP event pair per frame,
 * recorded on the launching stream right around the kernel, kept for up to RTGGX_KERNEL_RING frames; 3 like 2 for
 * every 8th frame only (an event pair costs the launching stream ~6 us per frame). */
#define RTGGX_KERNEL_RING 4096
int  rtggx_enable_timing(rtggx_context* ctx, int mode);
/* Durations (ms) of the ray-trace kernel launches recorded in mode 2 or 3 since the last call; synchronises. */
int  rtggx_kernel_times(rtggx_context* ctx, float* ms, uint32_t capacity, uint32_t* count);

/* Diagnostic: the shader clock (MHz) the device runs at while the call is in flight -- one idle wave on the refit stream compares the
 * shader-cycle counter with the 100 MHz real-time counter over ~20 us; other streams keep running.  Synchronises only that stream. */
int  rtggx_debug_shader_clock(rtggx_context* ctx, double* mhz);

/* Attainable HBM bandwidth of the device (GB/s, read + written bytes): a float4 copy kernel over two buffers of `bytes` each,
 * `iterations` timed launches for each of four launch shapes (2 / 4 / 8 / 16 workgroups per CU: the shape matters by 20 % on MI355X), the
 * best one reported.  For the measured peak bench.py quotes beside the vendor figure; synchronises. */
int  rtggx_copy_bandwidth(rtggx_context* ctx, size_t bytes, int iterations, double* gbytes_per_s);

/* Size in bytes of a buffer / synchronous copy into caller memory / raw device pointer. */
int  rtggx_buffer_size(rtggx_context* ctx, int buffer_id, size_t* bytes);
int  rtggx_readback(rtggx_context* ctx, int buffer_id, void* dst, size_t bytes);
int  rtggx_buffer_ptr(rtggx_context* ctx, int buffer_id, void** device_ptr);
/* Overwrite a render target from host memory (tests feed one pass with another implementation's input). */
int  rtggx_upload(rtggx_context* ctx, int buffer_id, const void* src, size_t bytes);
int  rtggx_frame_parity(rtggx_context* ctx, uint32_t* parity);
int  rtggx_bvh_root(rtggx_context* ctx, uint32_t slot, int32_t* root);

/* Closest-hit queries on the device for tests: rays = n x {o.xyz, d.xyz, tmin, tmax},
 * out = n x {t, instance(bits), primitive(bits), b1, b2, valid}. */
int  rtggx_trace_rays(rtggx_context* ctx, const float* rays, uint32_t n, float* out);

#ifdef __cplusplus
}
#endif
#endif /* RTGGX_H */
