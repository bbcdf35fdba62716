// C ABI of librtggx (include/rtggx.h): context management, scene upload, per-frame pass entry points.
// Frame order issued by the host (RayTracedGGX::OnRender, RayTracedGGX.cpp:302-353):
//   update_as (stream B)  ||  render_visibility (stream A)  -> event ->  ray_trace -> denoise -> tone_map
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
#include "rtggx_context.h"
#include <chrono>
#include "rt_queue.h"

#ifndef RT_REFIT_REBUILD_RATIO
#define RT_REFIT_REBUILD_RATIO 1.2f      // a refitted tree whose cost has grown by this factor since its build is rebuilt (rtggx_refit_as)
#endif
// The events that order the streams of one context among each other, and the frames-in-flight fence.  (Measured in round 4: with
// hipEventReleaseToDevice the 1080p frame gets SLOWER, 0.184 -> 0.191-0.197 ms; hipEventDisableSystemFence changes nothing.)
#define RT_EVENT_FLAGS (hipEventDisableTiming)
#ifndef RT_REBUILD_STEPS
#define RT_REBUILD_STEPS 16u             // launches of such a rebuild issued per frame (the bunny's build is ~75: five frames)
#endif
namespace rt {
static thread_local char g_err[512] = "";
void setError(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
}

__global__ void uploadParamsKernel(FrameParams src, FrameParams* dst) {
  // 912 bytes: one wave copies the by-value argument into the device-resident slot
  const uint32_t* s = reinterpret_cast<const uint32_t*>(&src);
  uint32_t* d = reinterpret_cast<uint32_t*>(dst);
  for (uint32_t i = threadIdx.x; i < sizeof(FrameParams) / 4; i += blockDim.x) d[i] = s[i];
}
__global__ void __launch_bounds__(256) copyKernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u) dst[i] = src[i];
}
typedef float __attribute__((ext_vector_type(4))) CopyVec4;
__global__ void __launch_bounds__(256) copyKernelNT(const CopyVec4* __restrict__ src, CopyVec4* __restrict__ dst, size_t n) {      // the data is used once: non-temporal loads and stores
  for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u) __builtin_nontemporal_store(__builtin_nontemporal_load(&src[i]), &dst[i]);
}
int uploadParams(rtggx_context* c, uint32_t slot, hipStream_t s) {
  hipLaunchKernelGGL(uploadParamsKernel, dim3(1), dim3(64), 0, s, c->slots[slot], c->dParams + slot);
  RT_HIP(hipGetLastError());
  return 0;
}
int uploadScene(rtggx_context* c, hipStream_t s) {
  Scene sc;
  for (int i = 0; i < 2; ++i) { sc.verts[i] = c->mesh[i].verts; sc.idx[i] = c->mesh[i].indices; sc.nodes[i] = c->mesh[i].nodes; sc.tris[i] = c->mesh[i].tris; sc.root[i] = c->mesh[i].root; }
  sc.env = c->env.texels; sc.envSize = c->env.size; sc.envMips = c->env.mips;
  for (int m = 0; m < 16; ++m) sc.mipOffset[m] = c->env.mipOffset[m];
  sc.sh = c->sh; sc.cosSin = c->cosSinTab;
  RT_HIP(hipMemcpyAsync(c->dScene, &sc, sizeof sc, hipMemcpyHostToDevice, s));
  RT_HIP(hipStreamSynchronize(s));
  c->sceneDirty = false;
  return 0;
}

// world -> object matrices of the two instances: general 4x4 inverse by cofactors in double,
// rounded once (the software TLAS; DESIGN.md "TLAS").
static void invert4x4(const float* a /*row-major*/, float* out) {
  double m[16], inv[16];
  for (int i = 0; i < 16; ++i) m[i] = (double)a[i];
  inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
  inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
  inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
  inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
  inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
  inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
  inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
  inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
  inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
  inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
  inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
  inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
  inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
  inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
  inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
  inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
  const double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
  const double rdet = 1.0 / det;
  for (int i = 0; i < 16; ++i) out[i] = (float)(inv[i] * rdet);
}

// The frames-in-flight fence (evSetRead[set]: the last reader of an input set has ended) rides on a kernel's completion signal, and a
// kernel that carries an event leaves its queue idle for ~5 us behind it (profiles/r03_c_strip_chain.txt).  So it rides on the LAST
// kernel the main stream gets for the frame: the fused temporal + tone-map kernel, or the tone map where that is a kernel of its own.
// A frame that ends earlier -- the caller traces without denoising -- gets the event recorded explicitly by the next frame (settleSetRead).
static void settleSetRead(rtggx_context* c) {
  if (c->setReadDeferred < 0) return;
  hipEventRecord(c->evSetRead[c->setReadDeferred], c->streamMain);
  c->setReadRecorded[c->setReadDeferred] = true; c->setReadDeferred = -1;
}
static hipError_t syncStreams(rtggx_context* c) {
  hipError_t e = c->ownVis ? hipStreamSynchronize(c->ownVis) : hipSuccess;
  if (e == hipSuccess && c->streamRefit) e = hipStreamSynchronize(c->streamRefit);
  if (e == hipSuccess) e = hipStreamSynchronize(c->ownAS);
  if (e == hipSuccess) e = hipStreamSynchronize(c->streamMain);
  return e;
}
// Constants reach the device in rtggx_update_as; a caller that skips it still gets them, on the main stream.
// The constants go up on stream B (which also runs the visibility pass); everything on the main stream that
// consumes them is ordered behind the event.
static int uploadParamsStreamB(rtggx_context* c) {
  // (a visibility pass that has carried this slot to the device already -- rtggx_update_as came after it -- writes the same words from
  // its first workgroup, on another stream: this upload, the newer TLAS, must land second)
  if (c->evVisStream && c->evVisStream != c->streamAS) RT_HIP(hipStreamWaitEvent(c->streamAS, c->evVis, 0));
  const int r = uploadParams(c, c->slot, c->streamAS);
  if (r) return r;
  c->slotUploaded = true;
  RT_HIP(hipEventRecord(c->evAS, c->streamAS));
  RT_HIP(hipStreamWaitEvent(c->streamMain, c->evAS, 0));
  return 0;
}
static int ensureParams(rtggx_context* c) { return c->slotUploaded ? 0 : uploadParamsStreamB(c); }
// Vertex storage of a mesh: one allocation aliased by all input sets, or one per set once the mesh deforms; the staging ring.
static void freeMeshVerts(MeshDev& m) {
  for (int i = 0; i < RT_SETS; ++i) { bool dup = false; for (int j = 0; j < i; ++j) dup = dup || m.vertsBuf[j] == m.vertsBuf[i]; if (!dup && m.vertsBuf[i]) hipFree(m.vertsBuf[i]); }
  for (auto& b : m.vertsBuf) b = nullptr;
  for (int i = 0; i < RT_SETS; ++i) { bool dup = false; for (int j = 0; j < i; ++j) dup = dup || m.fatBuf[j] == m.fatBuf[i]; if (!dup && m.fatBuf[i]) hipFree(m.fatBuf[i]); }
  for (auto& b : m.fatBuf) b = nullptr;
  m.fat = nullptr;
  for (auto& st : m.stage) { if (st) hipHostFree(st); st = nullptr; }
  for (auto& st : m.deviceStage) { if (st) hipFree(st); st = nullptr; }
  if (m.evProduced) { hipEventDestroy(m.evProduced); hipEventDestroy(m.evStaged); m.evProduced = m.evStaged = nullptr; }
  m.pendingDeviceStage = -1; m.deviceStageNext = 0;
  m.verts = nullptr; m.deforming = false; m.pendingStage = -1; m.version = 0; m.latestSet = 0;
  for (auto& v : m.vertsVersion) v = 0;
}
static int setMeshImpl(rtggx_context* c, uint32_t slot, const float* verts, uint32_t nv, const uint32_t* idx, uint32_t ni) {
  MeshDev& m = c->mesh[slot];
  RT_HIP(syncStreams(c));       // frames in flight on any of the streams still read the buffers freed below
  abandonRebuild(c, slot); m.wantRebuild = false;
  freeMeshVerts(m);
  freeBuildProducts(m);
  if (m.indices) { hipFree(m.indices); m.indices = nullptr; }
  m.root = -1; m.numVerts = nv; m.numIndices = ni; m.numTris = ni / 3;
  for (uint32_t i = 0; i < ni; ++i) if (idx[i] >= nv) { setError("rtggx_set_mesh: index %u out of range (%u vertices)", idx[i], nv); m.numVerts = m.numIndices = m.numTris = 0; return -1; }
  if (nv && ni) {
    RT_HIP(hipMalloc(&m.verts, sizeof(float) * 6 * (size_t)nv));
    for (auto& b : m.vertsBuf) b = m.verts;      // static until rtggx_refit_as: one allocation for all input sets
    RT_HIP(hipMalloc(&m.indices, sizeof(uint32_t) * (size_t)ni));
    RT_HIP(hipMemcpy(m.verts, verts, sizeof(float) * 6 * (size_t)nv, hipMemcpyHostToDevice));
    RT_HIP(hipMemcpy(m.indices, idx, sizeof(uint32_t) * (size_t)ni, hipMemcpyHostToDevice));
    for (int k = 0; k < 3; ++k) { m.bmin[k] = 3.4e38f; m.bmax[k] = -3.4e38f; }
    for (uint32_t v = 0; v < nv; ++v) for (int k = 0; k < 3; ++k) { const float x = verts[6 * (size_t)v + k]; if (x < m.bmin[k]) m.bmin[k] = x; if (x > m.bmax[k]) m.bmax[k] = x; }
    RT_HIP(hipMalloc(&m.fat, sizeof(float4) * 5 * (size_t)m.numTris));
    for (auto& b : m.fatBuf) b = m.fat;
    { const int r = buildFatTris(c, slot, 0, c->streamMain); if (r) return r; }
    RT_HIP(hipStreamSynchronize(c->streamMain));
  }
  c->asBuilt = false; c->sceneDirty = true;
  return 0;
}

static const float kGroundVerts[24][6] = {   // RayTracer::createGroundMesh, RayTracer.cpp:430-461
  {-1, 1, -1, 0, 1, 0}, {1, 1, -1, 0, 1, 0}, {1, 1, 1, 0, 1, 0}, {-1, 1, 1, 0, 1, 0},
  {-1, -1, -1, 0, -1, 0}, {1, -1, -1, 0, -1, 0}, {1, -1, 1, 0, -1, 0}, {-1, -1, 1, 0, -1, 0},
  {-1, -1, 1, -1, 0, 0}, {-1, -1, -1, -1, 0, 0}, {-1, 1, -1, -1, 0, 0}, {-1, 1, 1, -1, 0, 0},
  {1, -1, 1, 1, 0, 0}, {1, -1, -1, 1, 0, 0}, {1, 1, -1, 1, 0, 0}, {1, 1, 1, 1, 0, 0},
  {-1, -1, -1, 0, 0, -1}, {1, -1, -1, 0, 0, -1}, {1, 1, -1, 0, 0, -1}, {-1, 1, -1, 0, 0, -1},
  {-1, -1, 1, 0, 0, 1}, {1, -1, 1, 0, 0, 1}, {1, 1, 1, 0, 0, 1}, {-1, 1, 1, 0, 0, 1}};
static const uint32_t kGroundIdx[36] = {3, 1, 0, 2, 1, 3, 6, 4, 5, 7, 4, 6, 11, 9, 8, 10, 9, 11,   // :477-496
                                        14, 12, 13, 15, 12, 14, 19, 17, 16, 18, 17, 19, 22, 20, 21, 23, 20, 22};
}  // namespace rt

using namespace rt;

#define RT_CHECK_CTX(c) do { if (!(c)) { rt::setError("null context"); return -1; } hipError_t _e = hipSetDevice((c)->device); if (_e != hipSuccess) { rt::setError("hipSetDevice: %s", hipGetErrorString(_e)); return -2; } } while (0)

extern "C" {

const char* rtggx_last_error(void) { return rt::g_err; }

int rtggx_create(rtggx_context** out, uint32_t width, uint32_t height, int device) {
  if (!out || width == 0 || height == 0 || width > 16384 || height > 16384) { setError("rtggx_create: bad arguments"); return -1; }
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { setError("rtggx_create: no HIP device available (this library has no CPU path)"); return -2; }
  if (device < 0 || device >= count) { setError("rtggx_create: device %d out of range (%d devices)", device, count); return -1; }
  RT_HIP(hipSetDevice(device));
  rtggx_context* c = new rtggx_context();
  c->device = device; c->W = width; c->H = height; c->rowBegin = 0; c->rowEnd = height;
  const size_t n = (size_t)width * height;
  // Streams and priorities (measured in rounds 1-3, profiles/r02_c_ab_pipeline.txt; the switches that chose between them are gone):
  //   main  high   hit shading, spatial filters, temporal pass + tone map: the longest chain of the three, and the one the others slow down most
  //   B     low    the traversal (one resident workgroup per CU)
  //   C     low    visibility pass + ray generation of the next frame
  //   R     middle vertex upload + tree refit of a deforming mesh; the traversals of odd frames where launches are small
  int prioLeast = 0, prioGreatest = 0;
  RT_HIP(hipDeviceGetStreamPriorityRange(&prioLeast, &prioGreatest));
  const int prioMid = (prioLeast + prioGreatest) / 2;
  RT_HIP(hipStreamCreateWithPriority(&c->ownMain, hipStreamNonBlocking, prioGreatest));
  RT_HIP(hipStreamCreateWithPriority(&c->ownAS, hipStreamNonBlocking, prioLeast));
  c->streamMain = c->ownMain; c->streamAS = c->ownAS;
  c->attachEvents = !(getenv("RTGGX_ATTACH_EVENTS") && atoi(getenv("RTGGX_ATTACH_EVENTS")) == 0);      // 0: marker packets (hipEventRecord) instead of events riding on kernels
  RT_HIP(hipStreamCreateWithPriority(&c->ownVis, hipStreamNonBlocking, prioLeast)); c->streamVis = c->ownVis;
  RT_HIP(hipEventCreateWithFlags(&c->evVis, RT_EVENT_FLAGS));
  RT_HIP(hipEventCreateWithFlags(&c->evRefit, RT_EVENT_FLAGS));
  for (auto& e : c->evGenRing) RT_HIP(hipEventCreateWithFlags(&e, RT_EVENT_FLAGS));
  for (auto& e : c->evTraceRing) RT_HIP(hipEventCreateWithFlags(&e, RT_EVENT_FLAGS));
  RT_HIP(hipStreamCreateWithPriority(&c->streamRefit, hipStreamNonBlocking, prioMid));
  c->rebuildRatio = RT_REFIT_REBUILD_RATIO; c->rebuildSteps = RT_REBUILD_STEPS;      // rtggx_set_refit_policy
  RT_HIP(hipEventCreateWithFlags(&c->evAS, RT_EVENT_FLAGS));
  RT_HIP(hipEventCreateWithFlags(&c->evRT, RT_EVENT_FLAGS));
  for (auto& e : c->evSetRead) RT_HIP(hipEventCreateWithFlags(&e, RT_EVENT_FLAGS));
  for (auto& e : c->tev) RT_HIP(hipEventCreate(&e));
  for (int i = 0; i < RT_SETS; ++i) {
    RT_HIP(hipMalloc(&c->normalBuf[i], n * 4)); RT_HIP(hipMemset(c->normalBuf[i], 0, n * 4));
    RT_HIP(hipMalloc(&c->depth32Buf[i], n * 4)); RT_HIP(hipMemset(c->depth32Buf[i], 0, n * 4));
    RT_HIP(hipMalloc(&c->velocityBuf[i], n * 4)); RT_HIP(hipMemset(c->velocityBuf[i], 0, n * 4));
    RT_HIP(hipMalloc(&c->rtReflBuf[i], n * 4)); RT_HIP(hipMemset(c->rtReflBuf[i], 0, n * 4));
    RT_HIP(hipMalloc(&c->rtDiffBuf[i], n * 4)); RT_HIP(hipMemset(c->rtDiffBuf[i], 0, n * 4));
    RT_HIP(hipMalloc(&c->roughMetalBuf[i], n * 2)); RT_HIP(hipMemset(c->roughMetalBuf[i], 0, n * 2));
  }
  for (auto& b : c->visDepthBuf) { RT_HIP(hipMalloc(&b, n * 8)); RT_HIP(hipMemset(b, 0, n * 8)); }
  { const size_t tiles = (size_t)((width + 15) / 16) * ((height + 15) / 16 + 1);      // (+ a row: a strip's tiles start at its first row)
    for (auto& b : c->visDirtyBuf) { RT_HIP(hipMalloc(&b, tiles * 4)); RT_HIP(hipMemset(b, 0xFF, tiles * 4)); }
    RT_HIP(hipMalloc(&c->visDirtyOnes, tiles * 4)); RT_HIP(hipMemset(c->visDirtyOnes, 0xFF, tiles * 4)); }
  c->selectSet(0);
  RT_HIP(hipMalloc(&c->backbuffer, n * 4));
  RT_HIP(hipMalloc(&c->tss[0], n * 8)); RT_HIP(hipMalloc(&c->tss[1], n * 8)); RT_HIP(hipMalloc(&c->fltRfl, n * 8)); RT_HIP(hipMalloc(&c->fltDff, n * 8));
  RT_HIP(hipMemset(c->backbuffer, 0, n * 4));
  RT_HIP(hipMemset(c->tss[0], 0, n * 8)); RT_HIP(hipMemset(c->tss[1], 0, n * 8)); RT_HIP(hipMemset(c->fltRfl, 0, n * 8)); RT_HIP(hipMemset(c->fltDff, 0, n * 8));
  c->largeCapacity = 1u << 16;
  for (auto& b : c->largeTrisBuf) RT_HIP(hipMalloc(&b, (size_t)c->largeCapacity * 56));
  RT_HIP(hipMalloc(&c->largeCountBase, 4 * (2 + RT_SETS))); RT_HIP(hipMemset(c->largeCountBase, 0, 4 * (2 + RT_SETS)));
  RT_HIP(hipMalloc(&c->rayCounter, 512 * 8)); RT_HIP(hipMemset(c->rayCounter, 0, 512 * 8));
  RT_HIP(hipMalloc(&c->rayCounterBuf, 1792 * 4)); RT_HIP(hipMemset(c->rayCounterBuf, 0, 1792 * 4));      // [4][256] per-frame counters + 768 statistics words
  c->rayCounter32 = c->lastRayCounter32 = c->rayCounterBuf;
  RT_HIP(hipMalloc(&c->traceStamps, 8 * 8)); RT_HIP(hipMemset(c->traceStamps, 0, 8 * 8));
  RT_HIP(hipHostMalloc(&c->hostRayCounters, 264 * 4)); memset(c->hostRayCounters, 0, 264 * 4); RT_HIP(hipEventCreateWithFlags(&c->evRayCounters, hipEventDisableTiming));   // [0..255] rays; [256] split demand; [258..261] duration and period of a trace launch (two 64-bit words)
  {
    hipDeviceProp_t prop;
    RT_HIP(hipGetDeviceProperties(&prop, device));
    c->numCUs = prop.multiProcessorCount > 0 ? (uint32_t)prop.multiProcessorCount : 256u;
    // one bin of 128 ray slots per 8x8 pixel sub-tile (4 per 16x16 tile): at most 2 rays per pixel
    const uint32_t tiles = ((width + 15) / 16) * ((height + 15) / 16);
    c->numBinsMax = tiles * 4u;
    if (c->numBinsMax < 64u) c->numBinsMax = 64u;            // room for rtggx_trace_rays batches on tiny frames
    c->binSlots = RT_BIN_MIN;      // (all-metal default materials: one ray per pixel; rtggx_update_frame grows the bins with the first metallic below 1)
    for (int i = 0; i < RT_SETS; ++i) {
      RT_HIP(hipMalloc(&c->rayQueueBuf[i], (size_t)c->numBinsMax * c->binSlots * sizeof(rt::RayRec)));
      RT_HIP(hipMalloc(&c->hitQueueBuf[i], (size_t)c->numBinsMax * c->binSlots * 8));
      RT_HIP(hipMalloc(&c->binCountBuf[i], (size_t)c->numBinsMax * 4)); RT_HIP(hipMemset(c->binCountBuf[i], 0, (size_t)c->numBinsMax * 4));
    }
    c->selectSet(0);
    for (auto& b : c->binWorkBuf) { RT_HIP(hipMalloc(&b, (size_t)c->numBinsMax * 4)); RT_HIP(hipMemset(b, 0, (size_t)c->numBinsMax * 4)); }
    c->binWork = c->binWorkBuf[0];
    for (int i = 0; i < RT_SETS; ++i) RT_HIP(hipMalloc(&c->splitListBuf[i], (size_t)RT_SPLIT_CAP * 4));
    c->selectSet(0);
    c->splitWork = RT_SPLIT_WORK; c->splitMaxShift = RT_SPLIT_MAX_SHIFT;      // rtggx_debug_trace_split
    RT_HIP(hipMalloc(&c->dEnvMipOffset, 16 * 4)); RT_HIP(hipMemset(c->dEnvMipOffset, 0, 16 * 4));
    RT_HIP(hipMalloc(&c->dummyRecord, 128)); RT_HIP(hipMemset(c->dummyRecord, 0, 128));
  }
  RT_HIP(hipMalloc(&c->histReach, 4)); RT_HIP(hipMemset(c->histReach, 0, 4));
  RT_HIP(hipMalloc(&c->exchangeTokens, 4 * 2 * RT_MAX_PEERS)); RT_HIP(hipMemset(c->exchangeTokens, 0, 4 * 2 * RT_MAX_PEERS));
  RT_HIP(hipMalloc(&c->dPeerTable, sizeof(void*) * 2 * RT_MAX_PEERS + 4 * (RT_MAX_PEERS + 1))); RT_HIP(hipMemset(c->dPeerTable, 0, sizeof(void*) * 2 * RT_MAX_PEERS + 4 * (RT_MAX_PEERS + 1)));
  RT_HIP(hipMalloc(&c->sh, 27 * 4)); RT_HIP(hipMemset(c->sh, 0, 27 * 4));
  RT_HIP(hipMalloc(&c->cosSinTab, 512 * 4));
  RT_HIP(hipMalloc(&c->dParams, RT_SLOTS * sizeof(FrameParams)));
  RT_HIP(hipMalloc(&c->dScene, sizeof(Scene)));
  {  // cos/sin(2*pi*s/256): double libm, rounded once (RayTracing.hlsl:94,100 with xi.x = s/256, :391)
    float tab[512];
    for (int s = 0; s < 256; ++s) { const double phi = 2.0 * 3.14159265358979323846 * (double)s / 256.0; tab[s] = (float)cos(phi); tab[256 + s] = (float)sin(phi); }
    RT_HIP(hipMemcpy(c->cosSinTab, tab, sizeof tab, hipMemcpyHostToDevice));
  }
  // default materials, RayTracer.cpp:134-139
  const float bc0[4] = {0.95f, 0.93f, 0.88f, 1.0f}, bc1[4] = {1.0f, 0.71f, 0.29f, 1.0f};
  const float rm0[4] = {0.5f, 1.0f, 0.0f, 0.0f}, rm1[4] = {0.16f, 1.0f, 0.0f, 0.0f};
  memcpy(c->material.BaseColors[0], bc0, 16); memcpy(c->material.BaseColors[1], bc1, 16);
  memcpy(c->material.RoughMetals[0], rm0, 16); memcpy(c->material.RoughMetals[1], rm1, 16);
  memset(c->invWorld, 0, sizeof c->invWorld);
  for (int i = 0; i < 2; ++i) for (int k = 0; k < 4; ++k) c->invWorld[i][k * 5] = 1.0f;
  memset(c->slots, 0, sizeof c->slots);
  const int r = setMeshImpl(c, RTGGX_GROUND, &kGroundVerts[0][0], 24, kGroundIdx, 36);
  if (r) { return r; }
  // hipMemset of device memory does not wait on the host, and the null stream it runs on is not ordered against this context's
  // (non-blocking) streams: nothing of the first frame may overtake a clear
  RT_HIP(hipStreamSynchronize(nullptr));
  *out = c;
  return 0;
}

void rtggx_destroy(rtggx_context* c) {
  if (!c) return;
  hipSetDevice(c->device);
  hipDeviceSynchronize();
  for (void* p : c->ipcMapped) hipIpcCloseMemHandle(p);
  hipFree(c->dPeerTable); hipFree(c->exchangeTokens);
  for (auto& m : c->mesh) {
    freeMeshVerts(m); freeBuildProducts(m);
    hipFree(m.indices); hipFree(m.dCost); if (m.hCost) hipHostFree(m.hCost); if (m.evCost) hipEventDestroy(m.evCost);
  }
  hipFree(c->env.texels); hipFree(c->sh); hipFree(c->cosSinTab); hipFree(c->backbuffer);
  for (auto b : c->visDepthBuf) hipFree(b);
  for (auto b : c->visDirtyBuf) hipFree(b);
  hipFree(c->visDirtyOnes);
  for (int i = 0; i < RT_SETS; ++i) { hipFree(c->depth32Buf[i]); hipFree(c->normalBuf[i]); hipFree(c->velocityBuf[i]); hipFree(c->rtReflBuf[i]); hipFree(c->rtDiffBuf[i]); hipFree(c->roughMetalBuf[i]); }
  hipFree(c->tss[0]); hipFree(c->tss[1]);
  hipFree(c->fltRfl); hipFree(c->fltDff); hipFree(c->largeTrisBuf[0]); hipFree(c->largeTrisBuf[1]); hipFree(c->largeCountBase); hipFree(c->rayCounter); hipFree(c->dParams); hipFree(c->dScene);
  for (int i = 0; i < RT_SETS; ++i) { hipFree(c->rayQueueBuf[i]); hipFree(c->hitQueueBuf[i]); hipFree(c->binCountBuf[i]); }
  for (auto b : c->binWorkBuf) hipFree(b); for (int i = 0; i < RT_SETS; ++i) hipFree(c->splitListBuf[i]);
  hipFree(c->stackOverflow); hipFree(c->testRayRange); hipFree(c->dummyRecord); hipFree(c->histReach);
  hipFree(c->dEnvMipOffset); hipFree(c->rayCounterBuf); hipFree(c->traceStamps); hipHostFree(c->hostRayCounters); hipEventDestroy(c->evRayCounters);
  for (auto& e : c->kevBegin) hipEventDestroy(e);
  for (auto& e : c->kevEnd) hipEventDestroy(e);
  for (auto& e : c->tev) hipEventDestroy(e);
  hipEventDestroy(c->evAS); hipEventDestroy(c->evRT); for (auto e : c->evSetRead) hipEventDestroy(e);
  hipStreamDestroy(c->ownMain); hipStreamDestroy(c->ownAS); if (c->ownVis) hipStreamDestroy(c->ownVis);
  hipEventDestroy(c->evVis); hipEventDestroy(c->evRefit); for (auto e : c->evGenRing) hipEventDestroy(e); for (auto e : c->evTraceRing) hipEventDestroy(e);
  if (c->streamRefit) hipStreamDestroy(c->streamRefit);
  delete c;
}

int rtggx_set_strip(rtggx_context* c, uint32_t rowBegin, uint32_t rowEnd) {
  RT_CHECK_CTX(c);
  if (rowBegin > rowEnd || rowEnd > c->H) { setError("rtggx_set_strip: bad rows [%u,%u) for height %u", rowBegin, rowEnd, c->H); return -1; }
  c->rowBegin = rowBegin; c->rowEnd = rowEnd; c->toneMapDone = false;
  return 0;
}

// Multi-GPU strips: the caller exchanges `rows` rows of TemporalSSOut beyond each strip edge between frames.  The
// temporal pass reports reprojections that read further than that (rtggx_history_overreach): such frames differ from the
// single-GPU frame, and the caller widens the apron or reports it.
int rtggx_set_history_apron(rtggx_context* c, uint32_t rows) {
  RT_CHECK_CTX(c);
  c->historyApron = rows;
  return 0;
}
int rtggx_history_overreach(rtggx_context* c, uint32_t* rows, int reset) {
  RT_CHECK_CTX(c);
  if (!rows) { setError("rtggx_history_overreach: null result"); return -1; }
  RT_HIP(syncStreams(c));
  RT_HIP(hipMemcpy(rows, c->histReach, 4, hipMemcpyDeviceToHost));
  if (reset) { RT_HIP(hipMemset(c->histReach, 0, 4)); RT_HIP(hipStreamSynchronize(nullptr)); }
  return 0;
}

int rtggx_set_stream(rtggx_context* c, void* stream) {
  RT_CHECK_CTX(c);
  RT_HIP(syncStreams(c));
  if (stream) { c->streamMain = (hipStream_t)stream; c->externalStream = true; }
  else { c->streamMain = c->ownMain; c->externalStream = false; }
  if (!c->asyncCompute) c->streamAS = c->streamMain;
  return 0;
}

// Multi-GPU strips: the history images of ALL ranks, mapped into this process, for the temporal pass's taps beyond the exchanged apron
// ("N-strip output == 1-strip output on every buffer"; the reference samples its one history texture anywhere,
// CSTemporalSS.hlsl:259-265).  Every rank allocates full-size targets, so rank r's TemporalSSOut[p] holds the rows r owns at the same
// offsets as this rank's own image does; a tap at row y outside [b - apron, e + apron) is read from the image of the rank whose strip
// holds y.  `bounds`: world + 1 ascending rows (rank r owns [bounds[r], bounds[r + 1])); tss0 / tss1: world device pointers each, valid in
// THIS process -- another context's rtggx_buffer_ptr in the same process, or what rtggx_history_ipc_open returned for another process's
// rtggx_history_ipc_export.  The entries of this rank itself may be its own pointers or null.
// ORDERING is the caller's, and the per-frame exchange already provides it where it has a message in each direction between two ranks
// (include/rtggx.h): rank A's temporal pass of frame f + 1 may read rank B's image once B's temporal pass of frame f has ended (A's
// receive from B in the exchange of frame f), and B's H filter of frame f + 2, which reuses that image as scratch, waits for A's temporal
// pass of frame f + 1 (B's receive from A in the exchange of frame f + 1) -- both on the main streams the exchange is issued on.
int rtggx_set_history_peers(rtggx_context* c, uint32_t world, const uint32_t* bounds, void* const* tss0, void* const* tss1) {
  RT_CHECK_CTX(c);
  if (world == 0u) { RT_HIP(syncStreams(c)); c->peerWorld = 0u; return 0; }
  if (world > RT_MAX_PEERS || !bounds || !tss0 || !tss1) { setError("rtggx_set_history_peers: 1 .. %d ranks, boundaries and two pointer lists", RT_MAX_PEERS); return -1; }
  if (bounds[0] != 0u || bounds[world] != c->H) { setError("rtggx_set_history_peers: boundaries must run from 0 to the frame height %u", c->H); return -1; }
  for (uint32_t r = 0; r < world; ++r) if (bounds[r] > bounds[r + 1]) { setError("rtggx_set_history_peers: boundaries must ascend"); return -1; }
  RT_HIP(syncStreams(c));
  struct { const void* tss[2][RT_MAX_PEERS]; uint32_t bounds[RT_MAX_PEERS + 1]; } table;
  memset(&table, 0, sizeof table);
  for (uint32_t r = 0; r < world; ++r) {
    const bool own = bounds[r] <= c->rowBegin && c->rowEnd <= bounds[r + 1] && c->rowEnd > c->rowBegin;
    table.tss[0][r] = tss0[r] ? tss0[r] : own ? (void*)c->tss[0] : nullptr; table.tss[1][r] = tss1[r] ? tss1[r] : own ? (void*)c->tss[1] : nullptr;
    if (bounds[r + 1] > bounds[r] && (!table.tss[0][r] || !table.tss[1][r])) { setError("rtggx_set_history_peers: no history images for rank %u", r); return -1; }
  }
  for (uint32_t r = 0; r <= world; ++r) table.bounds[r] = bounds[r];
  for (uint32_t r = world + 1; r <= RT_MAX_PEERS; ++r) table.bounds[r] = c->H;
  static_assert(sizeof table == sizeof(void*) * 2 * RT_MAX_PEERS + 4 * (RT_MAX_PEERS + 1) + 4 || sizeof table == sizeof(void*) * 2 * RT_MAX_PEERS + 4 * (RT_MAX_PEERS + 1), "peer table layout");
  RT_HIP(hipMemcpy(c->dPeerTable, &table, sizeof(void*) * 2 * RT_MAX_PEERS + 4 * (RT_MAX_PEERS + 1), hipMemcpyHostToDevice));
  c->peerWorld = world;
  return 0;
}
// One process per GPU: the two history images as inter-process handles (2 x 64 bytes: hipIpcMemHandle_t of TemporalSSOut[0], [1]) ...
int rtggx_history_ipc_export(rtggx_context* c, void* handles, size_t bytes) {
  RT_CHECK_CTX(c);
  static_assert(sizeof(hipIpcMemHandle_t) == RTGGX_IPC_HANDLE_BYTES, "hipIpcMemHandle_t size");
  if (!handles || bytes < 2 * sizeof(hipIpcMemHandle_t)) { setError("rtggx_history_ipc_export: room for two %zu-byte handles", sizeof(hipIpcMemHandle_t)); return -1; }
  hipIpcMemHandle_t h[2];
  for (int p = 0; p < 2; ++p) RT_HIP(hipIpcGetMemHandle(&h[p], c->tss[p]));
  memcpy(handles, h, sizeof h);
  return 0;
}
// ... and another rank's handles opened in this process: two device pointers for rtggx_set_history_peers (unmapped by rtggx_destroy).
int rtggx_history_ipc_open(rtggx_context* c, const void* handles, size_t bytes, void** tss0, void** tss1) {
  RT_CHECK_CTX(c);
  if (!handles || bytes < 2 * sizeof(hipIpcMemHandle_t) || !tss0 || !tss1) { setError("rtggx_history_ipc_open: bad arguments"); return -1; }
  hipIpcMemHandle_t h[2]; memcpy(h, handles, sizeof h);
  void* p[2] = {nullptr, nullptr};
  for (int k = 0; k < 2; ++k) {
    RT_HIP(hipIpcOpenMemHandle(&p[k], h[k], hipIpcMemLazyEnablePeerAccess));
    c->ipcMapped.push_back(p[k]);
  }
  *tss0 = p[0]; *tss1 = p[1];
  return 0;
}

int rtggx_get_stream(rtggx_context* c, void** stream) {
  RT_CHECK_CTX(c);
  if (!stream) { setError("rtggx_get_stream: null"); return -1; }
  *stream = (void*)c->streamMain;
  return 0;
}

// The sample's [A] toggle / m_asyncCompute (RayTracedGGX.cpp:304-353 vs the single command list of :513-556).  Off: every
// pass of a frame is issued to ONE stream in submission order -- no stream B, no stream C, no overlap between the
// ray-tracing half of one frame and the denoising half of the previous one.  Results are identical either way.
int rtggx_set_async_compute(rtggx_context* c, int enable) {
  RT_CHECK_CTX(c);
  if ((enable != 0) == c->asyncCompute) return 0;
  RT_HIP(syncStreams(c));
  c->asyncCompute = enable != 0;
  c->streamAS = c->asyncCompute ? c->ownAS : c->streamMain;
  c->streamVis = c->asyncCompute ? c->ownVis : nullptr;
  c->evVisStream = nullptr; c->genStream = nullptr; for (auto& f : c->genFrame) f = 0u;
  return 0;
}

int rtggx_set_mesh(rtggx_context* c, uint32_t slot, const float* verts, uint32_t nv, const uint32_t* idx, uint32_t ni) {
  RT_CHECK_CTX(c);
  if (slot >= RTGGX_NUM_MESH || !verts || !idx || ni % 3 != 0) { setError("rtggx_set_mesh: bad arguments"); return -1; }
  return setMeshImpl(c, slot, verts, nv, idx, ni);
}

int rtggx_set_env(rtggx_context* c, int format, uint32_t size, uint32_t mips, const void* data, size_t bytes) {
  RT_CHECK_CTX(c);
  if (!data) { setError("rtggx_set_env: null data"); return -1; }
  RT_HIP(syncStreams(c));
  return decodeEnv(c, format, size, mips, data, bytes, c->streamMain);
}

int rtggx_set_material(rtggx_context* c, uint32_t mesh, const float baseColor[4], float roughness, float metallic) {
  RT_CHECK_CTX(c);
  if (mesh >= RTGGX_NUM_MESH) { setError("rtggx_set_material: bad mesh"); return -1; }
  memcpy(c->material.BaseColors[mesh], baseColor, 16);
  c->material.RoughMetals[mesh][0] = roughness; c->material.RoughMetals[mesh][1] = metallic;
  return 0;
}
int rtggx_set_sampler(rtggx_context* c, int vndf) {
  RT_CHECK_CTX(c);
  c->vndf = vndf != 0;      // takes effect with the next rtggx_update_frame
  return 0;
}
int rtggx_set_metallic(rtggx_context* c, uint32_t mesh, float metallic) {   // RayTracer.cpp:244-248
  RT_CHECK_CTX(c);
  if (mesh >= RTGGX_NUM_MESH) { setError("rtggx_set_metallic: bad mesh"); return -1; }
  c->material.RoughMetals[mesh][1] = metallic;
  return 0;
}

int rtggx_build_as(rtggx_context* c) {
  RT_CHECK_CTX(c);
  RT_HIP(syncStreams(c));
  // (a mesh that deforms keeps its per-set vertex buffers: it is built from its newest shape, and every input set gets the tree of
  // that set's own vertices -- lbvh.hip buildLbvh; a rebuild in progress beside the frames is dropped)
  for (uint32_t i = 0; i < 2; ++i) { abandonRebuild(c, i); c->mesh[i].wantRebuild = false; const int r = buildLbvh(c, i, c->streamAS); if (r) return r; }
  c->selectSet(c->setIndex);
  c->asBuilt = true; c->sceneDirty = true;
  return 0;
}

// Deforming meshes (the sample itself only turns a rigid instance, RayTracer.cpp:326-341): new vertices for an
// unchanged topology.  The call only STAGES them (one copy into pinned memory); the upload and the refit of the acceleration
// structure are issued by the next rtggx_render_visibility on stream R -- behind the previous frame's traversal, beside that
// frame's shading and denoising on the main stream -- without a synchronisation.  The topology stays the one the last build
// chose; when the tree's cost has grown by RT_REFIT_REBUILD_RATIO since that build, the mesh is REBUILT beside the frames (round 3;
// lbvh.hip startRebuild: a few launches per frame behind the frame's refit, the new topology swapped in between two frames).  This call
// never waits for the GPU -- except the first one for a mesh, which allocates the per-set buffers.
static int splitBvhPerSet(rtggx_context* c, uint32_t slot);
// The first new shape of a mesh: one vertex buffer per input set (rtggx_context.h), the staging ring, the tree arrays per set, the second
// topology and scratch memory of a rebuild beside the frames -- the one place where a refit call waits for the GPU.
static int beginDeforming(rtggx_context* c, uint32_t slot) {
  MeshDev& m = c->mesh[slot];
  if (m.deforming) return 0;
  const size_t bytes = sizeof(float) * 6 * (size_t)m.numVerts;
  RT_HIP(syncStreams(c));
  float* nv2[RT_SETS] = {}; float4* nf[RT_SETS] = {}; float* st[RT_SLOTS] = {};
  const size_t fb = sizeof(float4) * 5 * (size_t)m.numTris;
  bool ok = true;
  for (int i = 1; i < RT_SETS && ok; ++i) ok = hipMalloc(&nv2[i], bytes) == hipSuccess && hipMalloc(&nf[i], fb) == hipSuccess;
  for (int i = 0; i < RT_SLOTS && ok; ++i) ok = hipHostMalloc(&st[i], bytes) == hipSuccess;
  if (!ok) {       // nothing half-done is left behind
    for (int i = 1; i < RT_SETS; ++i) { hipFree(nv2[i]); hipFree(nf[i]); }
    for (auto p : st) if (p) hipHostFree(p);
    setError("rtggx_refit_as: out of memory for the per-set vertex buffers of mesh %u", slot); return -2;
  }
  for (int i = 1; i < RT_SETS; ++i) {
    m.vertsBuf[i] = nv2[i]; m.fatBuf[i] = nf[i];
    RT_HIP(hipMemcpy(m.vertsBuf[i], m.vertsBuf[0], bytes, hipMemcpyDeviceToDevice)); RT_HIP(hipMemcpy(m.fatBuf[i], m.fatBuf[0], fb, hipMemcpyDeviceToDevice));
  }
  for (int i = 0; i < RT_SLOTS; ++i) m.stage[i] = st[i];
  { const int r = splitBvhPerSet(c, slot); if (r) return r; }
  { const int r = prepareRebuild(c, slot); if (r) return r; }      // the second topology and the build's scratch memory: not in the frame loop
  // A device-to-device hipMemcpy does NOT wait on the host (only its issue is synchronous) and runs on the null stream, which the
  // non-blocking streams of this context are not ordered against: the next frame's refit on stream R wrote a set's nodes while the copy
  // of the OLD nodes into the same array was still on its way, and the copy landed last (found once GPU_MAX_HW_QUEUES=8 gave the null
  // stream a hardware queue of its own: test_deforming_mesh_async_refit_against_the_oracle, "child box must contain the child").
  RT_HIP(hipStreamSynchronize(nullptr));
  m.deforming = true; m.latestSet = c->setIndex;
  c->selectSet(c->setIndex);
  return 0;
}
static int checkRefit(rtggx_context* c, uint32_t slot, const void* verts, uint32_t nv, const char* who) {
  if (slot >= RTGGX_NUM_MESH || !verts) { setError("%s: bad arguments", who); return -1; }
  const MeshDev& m = c->mesh[slot];
  if (!c->asBuilt || !m.tris) { setError("%s: rtggx_build_as has not been called", who); return -1; }
  if (nv != m.numVerts) { setError("%s: %u vertices given, the mesh has %u (a new topology needs rtggx_set_mesh + rtggx_build_as)", who, nv, m.numVerts); return -1; }
  return 0;
}
// the cost of the tree after an earlier refit has arrived: has the shape drifted too far from the one the topology was built for?
static void pollTreeCost(rtggx_context* c, MeshDev& m) {
  if (m.costInFlight && hipEventQuery(m.evCost) == hipSuccess) { m.lastCost = *m.hCost; m.costInFlight = false; }
  if (m.builtCost > 0.0f && m.lastCost > c->rebuildRatio * m.builtCost) m.wantRebuild = true;      // started by the next frame (issuePendingRefits)
}
int rtggx_refit_as(rtggx_context* c, uint32_t slot, const float* verts, uint32_t nv) {
  RT_CHECK_CTX(c);
  { const int r = checkRefit(c, slot, verts, nv, "rtggx_refit_as"); if (r) return r; }
  { const int r = beginDeforming(c, slot); if (r) return r; }
  MeshDev& m = c->mesh[slot];
  pollTreeCost(c, m);
  // A shape that no frame has picked up yet (two calls between frames, calls without frames) is simply replaced: its staging buffer has
  // no copy in flight.  Otherwise the next buffer of the ring: it was last consumed RT_SLOTS frames ago, and the copy that read it was
  // ordered before a traversal whose frame the host has since waited for (the set fence of rtggx_render_visibility).
  uint32_t st;
  if (m.pendingStage >= 0) st = (uint32_t)m.pendingStage;
  else { st = m.stageNext; m.stageNext = (m.stageNext + 1u) % RT_SLOTS; }
  memcpy(m.stage[st], verts, sizeof(float) * 6 * (size_t)nv);
  m.pendingStage = (int)st; m.pendingDeviceStage = -1;      // (the newest shape wins)
  return 0;
}
// The same for a mesh that is animated ON the GPU (round 4): `dverts` is a device pointer, `stream` the stream that
// produces it (null: the null stream).  Like a hipMemcpyAsync on that stream: the copy out of `dverts` is ordered behind everything the
// stream holds at the time of the call, and what the caller gives the stream afterwards -- the next animation step, overwriting `dverts` --
// behind the copy.  It runs on the context's geometry stream, into a device-side staging ring the next frame's refit reads: no host copy
// (836 KB and 17 us per frame for the bunny through rtggx_refit_as), no wait.
int rtggx_refit_as_device(rtggx_context* c, uint32_t slot, const float* dverts, uint32_t nv, void* stream) {
  RT_CHECK_CTX(c);
  { const int r = checkRefit(c, slot, dverts, nv, "rtggx_refit_as_device"); if (r) return r; }
  { const int r = beginDeforming(c, slot); if (r) return r; }
  MeshDev& m = c->mesh[slot];
  const size_t bytes = sizeof(float) * 6 * (size_t)nv;
  if (!m.deviceStage[0]) {
    for (auto& p : m.deviceStage) RT_HIP(hipMalloc(&p, bytes));
    RT_HIP(hipEventCreateWithFlags(&m.evProduced, hipEventDisableTiming)); RT_HIP(hipEventCreateWithFlags(&m.evStaged, hipEventDisableTiming));
  }
  pollTreeCost(c, m);
  const hipStream_t s = c->asyncCompute ? c->streamRefit : c->streamMain, producer = (hipStream_t)stream;
  const uint32_t k = m.deviceStageNext; m.deviceStageNext = (m.deviceStageNext + 1u) % RT_SLOTS;      // (copies into the ring follow each other and their readers on one stream)
  if (producer != s) { RT_HIP(hipEventRecord(m.evProduced, producer)); RT_HIP(hipStreamWaitEvent(s, m.evProduced, 0)); }
  RT_HIP(hipMemcpyAsync(m.deviceStage[k], dverts, bytes, hipMemcpyDeviceToDevice, s));
  if (producer != s) { RT_HIP(hipEventRecord(m.evStaged, s)); RT_HIP(hipStreamWaitEvent(producer, m.evStaged, 0)); }
  m.pendingDeviceStage = (int)k; m.pendingStage = -1;
  return 0;
}
int rtggx_set_refit_policy(rtggx_context* c, float rebuildRatio, uint32_t stepsPerFrame) {
  RT_CHECK_CTX(c);
  if (!(rebuildRatio > 1.0f) || stepsPerFrame == 0u) { setError("rtggx_set_refit_policy: a ratio above 1 and at least one step per frame"); return -1; }
  c->rebuildRatio = rebuildRatio; c->rebuildSteps = stepsPerFrame;
  return 0;
}
int rtggx_refit_stats(rtggx_context* c, uint32_t slot, float* costRatio, uint32_t* refits, uint32_t* rebuilds) {
  RT_CHECK_CTX(c);
  if (slot >= RTGGX_NUM_MESH) { setError("rtggx_refit_stats: bad mesh"); return -1; }
  MeshDev& m = c->mesh[slot];
  RT_HIP(syncStreams(c));
  if (m.costInFlight) { m.lastCost = *m.hCost; m.costInFlight = false; }
  if (costRatio) *costRatio = m.builtCost > 0.0f ? m.lastCost / m.builtCost : 1.0f;
  if (refits) *refits = m.refits;
  if (rebuilds) *rebuilds = m.rebuilds;
  return 0;
}

// The boxes and leaf triangles of a deforming mesh once per input set (rtggx_context.h): copies of the tree just built.
static int splitBvhPerSet(rtggx_context* c, uint32_t slot) {
  MeshDev& m = c->mesh[slot];
  const size_t n = m.numTris, nn = n > 1 ? n - 1 : 1, topCap = slot == 0 ? RT_TOP_SLOT0 : RT_TOP_SLOT1;
  for (int i = 1; i < RT_SETS; ++i) {
    if (m.trisBuf[i] != m.trisBuf[0]) continue;
    RT_HIP(hipMalloc(&m.trisBuf[i], sizeof(BvhTri) * n)); RT_HIP(hipMemcpy(m.trisBuf[i], m.trisBuf[0], sizeof(BvhTri) * n, hipMemcpyDeviceToDevice));
    RT_HIP(hipMalloc(&m.nodesBuf[i], sizeof(BvhNode) * nn)); RT_HIP(hipMemcpy(m.nodesBuf[i], m.nodesBuf[0], sizeof(BvhNode) * nn, hipMemcpyDeviceToDevice));
    RT_HIP(hipMalloc(&m.nodes4Buf[i], sizeof(Bvh4Node) * nn)); RT_HIP(hipMemcpy(m.nodes4Buf[i], m.nodes4Buf[0], sizeof(Bvh4Node) * nn, hipMemcpyDeviceToDevice));
    RT_HIP(hipMalloc(&m.topBuf[i], sizeof(Bvh4Node) * topCap)); RT_HIP(hipMemcpy(m.topBuf[i], m.topBuf[0], sizeof(Bvh4Node) * topCap, hipMemcpyDeviceToDevice));
    m.topCountBuf[i] = m.topCountBuf[0];
  }
  return 0;
}

// Issued by rtggx_render_visibility once the new input set is selected and fenced: bring the set's vertex buffer up to date and
// refit -- everything on stream R; *touched: stream R was given work the visibility pass and the traversal must follow (evRefit, recorded
// HERE, right behind the refit).  A rebuild beside the frames gets its next launches only after that and after the visibility pass
// (issueRebuildSteps): they read the snapshot and write the job's own topology, nobody waits for them, and they must not sit between a
// frame's refit and what waits for it (round 3 issued them in front: while a rebuild was in progress every frame stalled behind 16
// build launches, the single-workgroup plocFinal among them).
static int issuePendingRefits(rtggx_context* c, bool* touched) {
  *touched = false;
  const hipStream_t s = c->asyncCompute ? c->streamRefit : c->streamMain;
  bool swapped[RTGGX_NUM_MESH] = {};
  for (uint32_t slot = 0; slot < RTGGX_NUM_MESH; ++slot) {
    MeshDev& m = c->mesh[slot];
    if (!m.deforming) continue;
    // has a rebuild whose launches are all out ended?  then this frame's refit is the first on the new topology
    { const int r = continueRebuild(c, slot, s, 0u, &swapped[slot]); if (r) return r; }
    const size_t bytes = sizeof(float) * 6 * (size_t)m.numVerts;
    const uint32_t set = c->setIndex;
    bool refit = true;
    if (m.pendingStage >= 0) {
      RT_HIP(hipMemcpyAsync(m.vertsBuf[set], m.stage[m.pendingStage], bytes, hipMemcpyHostToDevice, s));
      m.pendingStage = -1; ++m.version;
    } else if (m.pendingDeviceStage >= 0) {      // rtggx_refit_as_device: staged on this stream already
      RT_HIP(hipMemcpyAsync(m.vertsBuf[set], m.deviceStage[m.pendingDeviceStage], bytes, hipMemcpyDeviceToDevice, s));
      m.pendingDeviceStage = -1; ++m.version;
    } else if (m.vertsVersion[set] != m.version) {      // no new shape this frame: this set still holds an older one
      RT_HIP(hipMemcpyAsync(m.vertsBuf[set], m.vertsBuf[m.latestSet], bytes, hipMemcpyDeviceToDevice, s));
    } else refit = false;
    if (refit) {
      m.vertsVersion[set] = m.version; m.latestSet = set;
      const int r = refitLbvh(c, slot, set, s);           // this set's leaf triangles and nodes from this set's vertices
      if (r) return r;
      *touched = true;
    }
  }
  if (*touched && c->asyncCompute) RT_HIP(hipEventRecord(c->evRefit, s));
  for (uint32_t slot = 0; slot < RTGGX_NUM_MESH; ++slot) {
    MeshDev& m = c->mesh[slot];
    if (!m.deforming) continue;
    if (m.wantRebuild && !swapped[slot]) {      // (the cost that asked for it was the old topology's)
      const int r = startRebuild(c, slot, m.latestSet);
      if (r < 0) return r;
      if (r == 1) m.wantRebuild = false;
    } else if (swapped[slot]) m.wantRebuild = false;
  }
  return 0;
}
// ... and, once the frame's visibility pass (which follows the refit on the same stream where launches are full-size) is out as well:
// the next launches of a rebuild in progress (the first ones of one just started: the copy of the vertices it starts from).
static int issueRebuildSteps(rtggx_context* c) {
  const hipStream_t s = c->asyncCompute ? c->streamRefit : c->streamMain;
  for (uint32_t slot = 0; slot < RTGGX_NUM_MESH; ++slot) {
    if (!c->mesh[slot].deforming) continue;
    bool sw = false;
    const int r = continueRebuild(c, slot, s, c->rebuildSteps, &sw);
    if (r) return r;
  }
  return 0;
}

// A material with metallic below 1 traces a diffuse ray per covered pixel as well: two ray slots per pixel.  The bins grow once, before
// the first such frame (the frames in flight are waited for: the old bins are theirs).
static int growBins(rtggx_context* c) {
  RT_HIP(syncStreams(c));
  for (int i = 0; i < RT_SETS; ++i) {
    RT_HIP(hipFree(c->rayQueueBuf[i])); RT_HIP(hipFree(c->hitQueueBuf[i])); c->rayQueueBuf[i] = c->hitQueueBuf[i] = nullptr;
    RT_HIP(hipMalloc(&c->rayQueueBuf[i], (size_t)c->numBinsMax * RT_BIN * sizeof(rt::RayRec)));
    RT_HIP(hipMalloc(&c->hitQueueBuf[i], (size_t)c->numBinsMax * RT_BIN * 8));
  }
  if (c->testRayRange) { RT_HIP(hipFree(c->testRayRange)); c->testRayRange = nullptr; }
  c->binSlots = RT_BIN;
  c->selectSet(c->setIndex);
  return 0;
}
int rtggx_update_frame(rtggx_context* c, const RtggxFrameConstants* k) {
  RT_CHECK_CTX(c);
  if (!k) { setError("rtggx_update_frame: null constants"); return -1; }
  if (c->binSlots < RT_BIN && (c->material.RoughMetals[0][1] < 1.0f || c->material.RoughMetals[1][1] < 1.0f)) { const int r = growBins(c); if (r) return r; }
  c->slot = (c->slot + 1) % RT_SLOTS;   // RayTracer::FrameCount + 1 (rtggx_context.h)
  FrameParams& fp = c->slots[c->slot];
  fp.g = k->global; fp.rg = k->rayGen; fp.po[0] = k->perObject[0]; fp.po[1] = k->perObject[1];
  fp.mat = c->material;
  fp.W = c->W; fp.H = c->H; fp.rowBegin = c->rowBegin; fp.rowEnd = c->rowEnd;
  fp.flags = c->vndf ? RT_FLAG_VNDF : 0u; fp.pad[0] = fp.pad[1] = fp.pad[2] = 0u;
  memcpy(fp.invWorld, c->invWorld, sizeof fp.invWorld);
  c->haveConstants = true; c->slotUploaded = false;
  return 0;
}

// RayTracer::UpdateAccelerationStructure: refresh the two TLAS instance transforms from CBGlobal::Worlds
// (= m_worlds, RayTracer.cpp:288-290, 329-336).  Runs on the AS stream, overlapping the visibility pass.
int rtggx_update_as(rtggx_context* c) {
  RT_CHECK_CTX(c);
  if (!c->haveConstants) { setError("rtggx_update_as: rtggx_update_frame has not been called"); return -1; }
  FrameParams& fp = c->slots[c->slot];
  for (int i = 0; i < 2; ++i) {
    const M4 w = cbLoad4x3(fp.g.Worlds[i]);
    invert4x4(&w.m[0][0], c->invWorld[i]);
  }
  memcpy(fp.invWorld, c->invWorld, sizeof fp.invWorld);
  // Legal call order "render_visibility before update_as" (the sample's two queues overlap them, RayTracedGGX.cpp:304-339):
  // the visibility pass has then carried this slot to the device with the PREVIOUS frame's TLAS.  Mark it stale, so that
  // rtggx_ray_trace (ensureParams) sends it again, behind the visibility pass, before anything reads invWorld.
  c->slotUploaded = false;
  if (c->sceneDirty) { const int r = uploadScene(c, c->streamAS); if (r) return r; }
  // The constants (with the refreshed TLAS) ride to the device with the first kernel of the visibility pass, which
  // follows on stream B (rtggx_render_visibility); a caller that traces without a visibility pass gets them through
  // ensureParams.  In timing mode they are uploaded here, so that the pass has a duration of its own.
  if (c->timing) {
    hipEventRecord(c->tev[0], c->streamAS);
    const int r = uploadParamsStreamB(c);
    if (r) return r;
    hipEventRecord(c->tev[1], c->streamAS);
  }
  return 0;
}

int rtggx_transform_sh(rtggx_context* c) {
  RT_CHECK_CTX(c);
  return projectSH(c, c->streamAS);      // consumed by the shading kernel, which runs on stream B
}

// The frame on the device (RayTracedGGX::OnRender, RayTracedGGX.cpp:302-353, re-cut for this machine).  Three stages on three
// streams, each stage one frame behind the one before it:
//     stream C   visibility pass (its first kernel carries the frame constants) -> ray generation        of frame f + 1
//     stream B   traversal                                                                                of frame f
//     main       hit / miss shading -> spatial filters -> temporal pass + tone map                        of frame f - 1
// plus stream R for the vertex upload and tree refit of a deforming mesh.  No stage fills the machine by itself (the traversal
// is a latency-bound chain of dependent gathers: profiles/r02_*_limiter.txt), so the three overlap; what each stage hands to the next
// exists four times (the input sets), and the events are
//     evVis                  visibility f        -> ray generation f           (R -> C; stream order where both are on C)
//     evGenRing[f & 3]       ray generation f    -> traversal f                (C -> B)
//                            ray generation f    -> visibility f + 2           (C -> R: the target and the list it cleared)
//     evTraceRing[f & 3]     traversal f         -> shading f                  (B -> main)
//                            traversal f - 2     -> ray generation f           (B -> C: the bins' cost record and the ray counters
//                                                                               exist twice, by frame parity)
//     evRefit                refit f             -> visibility f, traversal f  (R -> C, B)
//     evSetRead[set]         last reader of a set -> the HOST, four frames later (the sample's frames-in-flight fence)
// rtggx_set_async_compute(0) (the sample's [A] toggle) puts everything on the main stream.
//
// WHERE a frame's kernels go is decided in ONE place, placeFrame, from a key of five facts (round 4: rounds 2-3 had grown nine
// interacting switches for it).  The table, each line measured in the round that introduced it (DESIGN.md sections 5, 7, 9):
//     key                              visibility pass   ray generation   traversal            hit shading       frames in flight
//     full-size launch                 C                 C                B                    main              4
//       + a mesh deforms / diffuse rays                                                                          3   (the front stages otherwise run ahead into one of two states)
//     small launch (< 200 000 rays)    C                 C                B, odd frames on R   the traversal's   4   (two traversals in flight; the main stream's chain is a strip's longest)
//       + a mesh deforms               C                 C                B                    main              4   (R is the refit's)
//     strip / caller-owned stream      no line of their own: rows and the main stream's identity do not move a kernel
//     async compute off                main              main             main                 main              4
struct Placement {
  bool small, strip, deforming, diffuse, callerStream;      // the key
  hipStream_t raster, gen, trace, shade;
  uint32_t framesInFlight;
  bool alternate, shadeWithTrace;
};
static Placement placeFrame(const rtggx_context* c, const FrameParams& fp, uint32_t frame) {
  Placement P;
  P.small = c->lastTraceSmall;      // by the ray count of the most recent frame whose count has arrived (raytrace.hip launchRayTrace; rtggx_debug_placement)
  P.strip = fp.rowBegin > 0u || fp.rowEnd < fp.H;
  P.deforming = c->mesh[0].deforming || c->mesh[1].deforming;
  P.diffuse = fp.mat.RoughMetals[0][1] < 1.0f || fp.mat.RoughMetals[1][1] < 1.0f;
  P.callerStream = c->externalStream;
  const bool async = c->asyncCompute && c->streamVis != nullptr;
  P.gen = async ? c->streamVis : c->streamMain;
  // (the visibility pass on the geometry stream R, beside the previous frame's ray generation instead of behind it -- built and measured in
  // round 4: 1080p 0.186 -> 0.202-0.216 ms, the traversal stretched from 0.146 to 0.226 ms by the busier mid-priority stream; profiles/r04_c_pipeline_ab.txt.
  // What it needed stays: ray generation clears the target of frame f + 2, and the pass waits for that ray generation by event.)
  P.raster = P.gen;
  P.alternate = async && c->streamRefit != nullptr && P.small && !P.deforming && (frame & 1u) != 0u;
  P.trace = !async ? c->streamMain : P.alternate ? c->streamRefit : c->streamAS;
  P.shadeWithTrace = async && c->attachEvents && !c->timing && P.small && !P.deforming;
  P.shade = P.shadeWithTrace ? P.trace : c->streamMain;
  P.framesInFlight = async && !P.small && (P.deforming || P.diffuse) ? RT_SETS - 1u : RT_SETS;
  return P;
}

static int waitForSet(rtggx_context* c, uint32_t set) {
  if (c->setReadRecorded[set] && hipEventQuery(c->evSetRead[set]) != hipSuccess) {
    const auto t0 = std::chrono::steady_clock::now();
    RT_HIP(hipEventSynchronize(c->evSetRead[set]));
    c->fenceWaitUs += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); ++c->fenceWaits;      // rtggx_debug_fence_wait
  }
  return 0;
}

int rtggx_render_visibility(rtggx_context* c) {
  RT_CHECK_CTX(c);
  if (!c->haveConstants) { setError("rtggx_render_visibility: no frame constants"); return -1; }
  if (!c->shDone && c->env.texels) { const int r = projectSH(c, c->streamAS); if (r) return r; }   // first frame only, RayTracer.cpp:345-350
  settleSetRead(c);      // (the previous frame ended without the kernel that would have carried its set's event)
  ++c->frameCounter;
  c->denoiseIssued = false; c->toneMapDone = false;
  c->selectSet((c->setIndex + 1u) % RT_SETS);
  // the set was last read four frames ago: normally long done; a host that has run further ahead than that waits here (also what makes
  // it safe for this frame's ray generation to clear the NEXT frame's visibility target: rtggx_context.h RT_VIS_RING)
  { const int r = waitForSet(c, c->setIndex); if (r) return r; }
  const Placement P = placeFrame(c, c->slots[c->slot], c->frameCounter);
  // THREE frames in flight where the table says so: the host also waits for the end of frame f - 3 (profiles/r03_i_deform_states.txt:
  // with four, the deforming bunny at 1080p ran at 0.212-0.219 or 0.26-0.30 ms per frame, a run fell into one state; with three 0.220-0.238)
  if (P.framesInFlight < RT_SETS) { const int r = waitForSet(c, (c->setIndex + RT_SETS - P.framesInFlight) % RT_SETS); if (r) return r; }
  c->refitIssued = false;
  { const int r = issuePendingRefits(c, &c->refitIssued); if (r) return r; }
  const hipStream_t s = P.raster;
  if (c->evVisStream && c->evVisStream != s) RT_HIP(hipStreamWaitEvent(s, c->evVis, 0));      // the previous pass ran on another stream
  // this frame's target and list of large triangles were cleared by the ray generation two frames back: on another stream, mostly
  { const uint32_t k = (c->frameCounter + 2u) & 3u;
    if (c->frameCounter >= 2u && c->genFrame[k] == c->frameCounter - 2u && c->genStreamOf[k] != s) RT_HIP(hipStreamWaitEvent(s, c->evGenRing[k], 0)); }
  // constants already on their way on stream B (timing mode uploads them in rtggx_update_as): the pass reads dParams[slot] and has to be
  // ordered behind that upload (evAS); on stream B it follows it anyway
  if (c->slotUploaded && s != c->streamAS) RT_HIP(hipStreamWaitEvent(s, c->evAS, 0));
  if (c->refitIssued && c->asyncCompute && s != c->streamRefit) RT_HIP(hipStreamWaitEvent(s, c->evRefit, 0));       // the rasteriser reads this set's vertices (on R it follows the refit anyway)
  if (c->timing) hipEventRecord(c->tev[2], s);
  int r = launchVisibility(c, c->slots[c->slot], s, c->streamVis ? c->evVis : nullptr);
  if (c->streamVis) c->evVisStream = s;
  if (c->timing) hipEventRecord(c->tev[13], s);
  c->genStream = P.gen;
  if (!r) r = issueRebuildSteps(c);
  return r;
}

int rtggx_ray_trace(rtggx_context* c) {
  RT_CHECK_CTX(c);
  if (!c->haveConstants || !c->asBuilt) { setError("rtggx_ray_trace: %s", c->asBuilt ? "no frame constants" : "rtggx_build_as has not been called"); return -1; }
  if (!c->env.texels) { setError("rtggx_ray_trace: no environment map"); return -1; }
  if (c->sceneDirty) { const int r = uploadScene(c, c->streamAS); if (r) return r; }
  { const int r = ensureParams(c); if (r) return r; }
  const uint32_t f = c->frameCounter;
  const FrameParams& fp = c->slots[c->slot];
  const Placement P = placeFrame(c, fp, f);
  c->lastPlacement[0] = P.small | (P.strip << 1) | (P.deforming << 2) | (P.diffuse << 3) | (P.callerStream << 4);
  const auto streamId = [&](hipStream_t st) { return st == c->streamMain ? 0u : st == c->streamAS ? 1u : st == c->streamVis ? 2u : st == c->streamRefit ? 3u : 4u; };
  c->lastPlacement[1] = streamId(P.gen) | (streamId(P.trace) << 4) | (streamId(P.shade) << 8) | (P.framesInFlight << 12) | (streamId(P.raster) << 16);
  const hipStream_t sGen = P.gen;
  if (c->evVisStream && c->evVisStream != sGen) RT_HIP(hipStreamWaitEvent(sGen, c->evVis, 0));      // this frame's visibility pass (on the geometry stream)
  // Small launches last as long as their longest chain of dependent traversal steps and leave most of the chip idle meanwhile: the
  // traversals of odd frames go to a second stream (R, idle unless a mesh deforms), so that two are in flight.  Everything a traversal
  // shares with its neighbours in time is per input set, per frame parity or per frame & 3, and everybody who needs its results waits
  // for its event, not for its stream; the stack spill area exists twice (launchTrace).  (A third traversal stream loses everywhere:
  // beyond four streams with work on them the queues take turns; profiles/r03_c_strip_chain.txt.)
  const hipStream_t sTrace = P.trace;
  c->traceSpillHalf = P.alternate ? 1u : 0u;
  if (sGen != sTrace) {
    // ray generation reads the cost record of the traversal two frames back and resets that frame's ray counters (frame parity)
    if (c->traceRecorded[(f + 2u) & 3u]) RT_HIP(hipStreamWaitEvent(sGen, c->evTraceRing[(f + 2u) & 3u], 0));
    // a caller that skipped the visibility pass (or uploaded constants on stream B): order ray generation behind the upload
    if (c->slotUploaded) RT_HIP(hipStreamWaitEvent(sGen, c->evAS, 0));
  }
  if (c->refitIssued && c->asyncCompute) RT_HIP(hipStreamWaitEvent(sTrace, c->evRefit, 0));      // this set's tree
  if (c->timing) hipEventRecord(c->tev[3], sGen);
  hipEvent_t evDone = c->evTraceRing[f & 3u];
  const bool shadeWithTrace = P.shadeWithTrace && sGen != sTrace && sTrace != c->streamMain;
  // who carries RayTracingOut1 over from the previous set where this frame traces no diffuse ray (raytrace.hip launchShade): ray
  // generation, unless the previous frame's shading kernel wrote into that set's image -- then, once, ray generation waits for it
  // (one bubble in the pipeline), so that the previous set's image is final when it reads it
  if (c->shadeWroteDiff && !P.diffuse && !c->lastFrameDiffuse && c->shadeStream) {
    if (c->shadeStream != sGen) { RT_HIP(hipEventRecord(c->evRT, c->shadeStream)); RT_HIP(hipStreamWaitEvent(sGen, c->evRT, 0)); }
    c->shadeWroteDiff = false;
  }
  c->lastFrameDiffuse = P.diffuse;
  c->genCarriesDiff = !c->shadeWroteDiff;
  int r = launchRayTrace(c, fp, sGen, sTrace, shadeWithTrace ? nullptr : evDone);
  c->traceRecorded[f & 3u] = true;
  c->lastRayCounter32 = c->rayCounter32;
  c->shadeWroteDiff = !c->genCarriesDiff || P.diffuse;
  if (shadeWithTrace) {
    // Small launches: the hit shading follows the traversal on ITS stream (the main stream's chain -- shading, two filters, temporal pass +
    // tone map -- is the longest stage of a thin strip's frame, and the two traversal streams alternate, so theirs may be twice as long:
    // 1920 x 171 0.060 -> 0.052 ms per frame, profiles/r03_c_strip_chain.txt); the event the main stream -- and ray generation two frames
    // on -- waits for then rides on the shading kernel.  The shading of frame f copies what it does not trace from the image of frame
    // f - 1, which the other traversal stream's shading kernel wrote, or the main stream's if this is the first frame shaded here.
    if (!c->genCarriesDiff && c->shadeStream && c->shadeStream != sTrace) {
      if (c->shadeStream == c->streamMain) { RT_HIP(hipEventRecord(c->evRT, c->streamMain)); RT_HIP(hipStreamWaitEvent(sTrace, c->evRT, 0)); }
      else if (c->traceRecorded[(f + 3u) & 3u]) RT_HIP(hipStreamWaitEvent(sTrace, c->evTraceRing[(f + 3u) & 3u], 0));
    }
    if (!r) r = launchShade(c, fp, sTrace, evDone);
    c->shadeStream = sTrace;
    RT_HIP(hipStreamWaitEvent(c->streamMain, evDone, 0));
  } else {
    // stream B runs ahead with the traversal; shading and the denoiser consume the bins, the G-buffer and the traced images on the main
    // stream (the event completes with the trace kernel; a shading kernel of the frame before on a traversal stream has been waited for
    // by the main stream in its own frame)
    RT_HIP(hipStreamWaitEvent(c->streamMain, evDone, 0));
    if (!r) r = launchShade(c, fp, c->streamMain, c->attachEvents ? nullptr : c->evSetRead[c->setIndex]);
    c->shadeStream = c->streamMain;
  }
  // the main stream has now been given work that reads the current input set: that set may not be overwritten (four frames from now)
  // before evSetRead, which rides on the LAST kernel the main stream gets for this frame (settleSetRead)
  if (c->attachEvents) c->setReadDeferred = (int)c->setIndex; else { c->setReadRecorded[c->setIndex] = true; c->setReadDeferred = -1; }
  if (c->timing) hipEventRecord(c->tev[14], c->streamMain);
  return r;
}

int rtggx_denoise(rtggx_context* c, int useSharedMem) {
  RT_CHECK_CTX(c);
  if (!c->haveConstants) { setError("rtggx_denoise: no frame constants"); return -1; }
  if (c->timing) hipEventRecord(c->tev[9], c->streamMain);   // start of denoise
  // Denoiser::Denoise and Denoiser::ToneMap follow each other in every frame of the sample (RayTracedGGX.cpp:341-350), and the temporal pass can
  // tone-map its result as well (denoise.hip temporalToneKernel; rtggx_tone_map then finds its work done): one launch, one event-carrying
  // gap and 8 bytes per pixel less.  WHERE that pays was measured (profiles/r04_c_pipeline_ab.txt): on small launches -- thin strips, small
  // frames, bound by the host's launches and the main stream's chain of short kernels -- and NOT on full-size frames, where the fused
  // kernel's workgroups of 1024 threads and 45 KB of LDS find room on a CU shared with the traversal's resident workgroup and ray
  // generation later than four small ones do (1080p 0.186 -> 0.204 ms; 512 threads: 0.199).  So it follows the placement's `small`; not in
  // the per-pass timing mode (the tone map keeps a duration of its own); rtggx_debug_fuse_tone_map pins it either way.
  const bool fuse = !c->timing && (c->fuseToneMap > 0 || (c->fuseToneMap < 0 && c->lastTraceSmall));
  const bool carry = fuse || !c->attachEvents;      // the frame's last kernel on this stream carries the set's event; else the tone map will
  const int r = launchDenoise(c, c->slots[c->slot], useSharedMem, c->streamMain, carry ? c->evSetRead[c->setIndex] : nullptr, fuse);
  if (carry) { c->setReadRecorded[c->setIndex] = true; c->setReadDeferred = -1; } else c->setReadDeferred = (int)c->setIndex;
  c->denoiseIssued = true; c->toneMapDone = fuse && c->slots[c->slot].rowEnd > c->slots[c->slot].rowBegin;
  return r;
}

int rtggx_tone_map(rtggx_context* c) {
  RT_CHECK_CTX(c);
  if (!c->haveConstants) { setError("rtggx_tone_map: no frame constants"); return -1; }
  int r = 0;
  if (c->toneMapDone) c->toneMapDone = false;      // this frame's rtggx_denoise wrote the back buffer as well
  else {
    const bool carry = c->setReadDeferred >= 0 && c->attachEvents;
    r = launchToneMap(c, c->slots[c->slot], c->streamMain, carry ? c->evSetRead[c->setReadDeferred] : nullptr);
    if (carry && c->slots[c->slot].rowEnd > c->slots[c->slot].rowBegin) { c->setReadRecorded[c->setReadDeferred] = true; c->setReadDeferred = -1; }
  }
  settleSetRead(c);
  if (c->timing) { hipEventRecord(c->tev[10], c->streamMain); c->timingsPending = true; }
  return r;
}

int rtggx_sync(rtggx_context* c) {
  RT_CHECK_CTX(c);
  RT_HIP(syncStreams(c));
  return 0;
}

int rtggx_ray_count(rtggx_context* c, uint64_t* rays) {
  RT_CHECK_CTX(c);
  uint32_t h[256];
  RT_HIP(syncStreams(c));
  RT_HIP(hipMemcpy(h, c->lastRayCounter32, sizeof h, hipMemcpyDeviceToHost));
  uint64_t s = 0; for (auto v : h) s += v;
  *rays = s;
  return 0;
}

// Diagnostic counters of builds compiled with -DRT_TRACE_STATS (zero otherwise): lane node steps, lane leaf
// steps, wave iterations, refills -- accumulated since the last reset.
int rtggx_debug_counters(rtggx_context* c, uint32_t* out, uint32_t n, int reset) {
  RT_CHECK_CTX(c);
  if (n > 768) { setError("rtggx_debug_counters: at most 768 words"); return -1; }
  RT_HIP(syncStreams(c));
  RT_HIP(hipMemcpy(out, c->rayCounterBuf + 1024, (size_t)n * 4, hipMemcpyDeviceToHost));
  if (reset) { RT_HIP(hipMemset(c->rayCounterBuf + 1024, 0, 768 * 4)); RT_HIP(hipStreamSynchronize(nullptr)); }
  return 0;
}

int rtggx_debug_fence_wait(rtggx_context* c, double* usTotal, uint32_t* waits, int reset) {
  RT_CHECK_CTX(c);
  if (usTotal) *usTotal = c->fenceWaitUs;
  if (waits) *waits = c->fenceWaits;
  if (reset) { c->fenceWaitUs = 0.0; c->fenceWaits = 0u; }
  return 0;
}
int rtggx_debug_fuse_tone_map(rtggx_context* c, int mode) {
  RT_CHECK_CTX(c);
  c->fuseToneMap = mode < 0 ? -1 : mode != 0;      // from the next rtggx_denoise on
  return 0;
}
// force_small: -1 by the ray count, 0 / 1 the placement of a full-size / small launch whatever the count (from the next frame on).
// key / where (either may be null): the most recent rtggx_ray_trace's key (bit 0 small, 1 strip, 2 deforming, 3 diffuse, 4 caller-owned
// stream) and placement (bits 0-3 / 4-7 / 8-11: the streams of ray generation / traversal / hit shading -- 0 main, 1 B, 2 C, 3 R --,
// bits 12-15 frames in flight).
int rtggx_debug_tile_words(rtggx_context* c, int enable) {
  RT_CHECK_CTX(c);
  c->useTileWords = enable != 0;
  return 0;
}
int rtggx_debug_placement(rtggx_context* c, int forceSmall, uint32_t* key, uint32_t* where) {
  RT_CHECK_CTX(c);
  if (forceSmall < -1 || forceSmall > 1) { setError("rtggx_debug_placement: force_small is -1, 0 or 1"); return -1; }
  c->forcePlacement = forceSmall;
  if (forceSmall >= 0) c->lastTraceSmall = forceSmall == 1;
  if (key) *key = c->lastPlacement[0];
  if (where) *where = c->lastPlacement[1];
  return 0;
}
int rtggx_debug_collapse_weights(rtggx_context* c, const float* set, float* get) {
  RT_CHECK_CTX(c);
  if (set) { if (!(set[0] >= 0.0f) || !(set[1] >= 0.0f) || !(set[0] + set[1] > 0.0f)) { setError("rtggx_debug_collapse_weights: two non-negative weights, not both zero"); return -1; }
             c->collapseWeights[0] = set[0]; c->collapseWeights[1] = set[1]; }
  if (get) { get[0] = c->collapseWeights[0]; get[1] = c->collapseWeights[1]; }
  return 0;
}
int rtggx_debug_trace_residency(rtggx_context* c, uint32_t forceWaves, uint32_t* waves, float* share) {
  RT_CHECK_CTX(c);
  if (forceWaves != 0u && forceWaves != 10u && forceWaves != 12u && forceWaves != 14u && forceWaves != 16u) { setError("rtggx_debug_trace_residency: %u waves: 0, 10, 12, 14 or 16", forceWaves); return -1; }
  c->traceWavesForced = forceWaves;
  if (forceWaves) c->traceWaves = forceWaves;
  if (waves) *waves = c->traceWaves;
  if (share) *share = c->traceShare;
  return 0;
}

int rtggx_debug_trace_split(rtggx_context* c, uint32_t workPerWave, uint32_t maxShift, int capacity, uint32_t* lastDemand) {
  RT_CHECK_CTX(c);
  if (maxShift > 3u) { setError("rtggx_debug_trace_split: max_shift %u > 3", maxShift); return -1; }
  if (capacity > (int)RT_SPLIT_CAP) { setError("rtggx_debug_trace_split: capacity %d > %u", capacity, RT_SPLIT_CAP); return -1; }
  RT_HIP(syncStreams(c));
  if (lastDemand) RT_HIP(hipMemcpy(lastDemand, c->splitCount, 4, hipMemcpyDeviceToHost));
  c->splitWork = workPerWave; c->splitMaxShift = maxShift;
  c->splitCapForced = capacity < 0 ? 0xFFFFFFFFu : ((uint32_t)capacity / 32u) * 32u;
  return 0;
}

int rtggx_ray_total(rtggx_context* c, uint64_t* rays, int reset) {
  RT_CHECK_CTX(c);
  unsigned long long h[256];
  RT_HIP(syncStreams(c));
  RT_HIP(hipMemcpy(h, c->rayCounter + 256, sizeof h, hipMemcpyDeviceToHost));
  uint64_t s = 0; for (auto v : h) s += v;
  *rays = s;
  if (reset) { RT_HIP(hipMemset(c->rayCounter + 256, 0, sizeof h)); RT_HIP(hipStreamSynchronize(nullptr)); }
  return 0;
}

// mode 0: off; 1: every pass (rtggx_get_timings); 2: only the ray-trace kernel, one event pair per frame
// kept in a ring of `RTGGX_KERNEL_RING` frames (rtggx_kernel_times) -- no host synchronisation per frame.
int rtggx_enable_timing(rtggx_context* c, int mode) {
  RT_CHECK_CTX(c);
  c->timing = mode == 1; c->timingsPending = false;
  c->kernelRing = mode == 2 || mode == 3; c->kevCount = 0; c->ringStride = mode == 3 ? 8u : 1u; c->ringTick = 0;
  if (c->kernelRing && c->kevBegin.empty()) {
    c->kevBegin.resize(RTGGX_KERNEL_RING); c->kevEnd.resize(RTGGX_KERNEL_RING);
    for (uint32_t i = 0; i < RTGGX_KERNEL_RING; ++i) { RT_HIP(hipEventCreate(&c->kevBegin[i])); RT_HIP(hipEventCreate(&c->kevEnd[i])); }
  }
  return 0;
}
int rtggx_kernel_times(rtggx_context* c, float* ms, uint32_t capacity, uint32_t* count) {
  RT_CHECK_CTX(c);
  RT_HIP(syncStreams(c));
  const uint32_t n = c->kevCount < capacity ? c->kevCount : capacity;
  for (uint32_t i = 0; i < n; ++i) RT_HIP(hipEventElapsedTime(&ms[i], c->kevBegin[i], c->kevEnd[i]));
  *count = n;
  c->kevCount = 0;
  return 0;
}
int rtggx_get_timings(rtggx_context* c, RtggxTimings* out) {
  RT_CHECK_CTX(c);
  if (!c->timing || !c->timingsPending) { setError("rtggx_get_timings: timing not enabled or no complete frame"); return -1; }
  RT_HIP(syncStreams(c));
  auto ms = [&](int a, int b) { float t = 0.0f; hipEventElapsedTime(&t, c->tev[a], c->tev[b]); return t; };
  RtggxTimings t;
  t.update_as = ms(0, 1); t.visibility = ms(2, 13); t.ray_trace = ms(3, 14); t.spatial_refl_h = ms(9, 4); t.spatial_refl_v = ms(4, 5);
  t.spatial_diff_h = ms(5, 6); t.spatial_diff_v = ms(6, 7); t.temporal = ms(7, 8); t.tone_map = ms(8, 10); t.frame = ms(2, 10);
  t.ray_trace_kernel = ms(11, 12);
  *out = t; c->lastTimings = t;
  return 0;
}

static int bufferInfo(rtggx_context* c, int id, void** ptr, size_t* bytes) {
  const size_t n = (size_t)c->W * c->H;
  switch (id) {
    case RTGGX_BUF_VISIBILITY: case RTGGX_BUF_DEPTH: *ptr = nullptr; *bytes = n * 4; return 0;   // halves of visDepth: staged
    case RTGGX_BUF_NORMAL: *ptr = c->normal; *bytes = n * 4; return 0;
    case RTGGX_BUF_ROUGH_METAL: *ptr = c->roughMetal; *bytes = n * 2; return 0;
    case RTGGX_BUF_VELOCITY: *ptr = c->velocity; *bytes = n * 4; return 0;
    case RTGGX_BUF_RT_REFL: *ptr = c->rtRefl; *bytes = n * 4; return 0;
    case RTGGX_BUF_RT_DIFF: *ptr = c->rtDiff; *bytes = n * 4; return 0;
    case RTGGX_BUF_TSS0: *ptr = c->tss[0]; *bytes = n * 8; return 0;
    case RTGGX_BUF_TSS1: *ptr = c->tss[1]; *bytes = n * 8; return 0;
    case RTGGX_BUF_FLT_RFL: *ptr = c->fltRflIsFltDff ? c->fltDff : c->fltRfl; *bytes = n * 8; return 0;      // identical images when no diffuse pass ran: only one was written (denoise.hip launchDenoise)
    case RTGGX_BUF_FLT_DFF: *ptr = c->fltDff; *bytes = n * 8; return 0;
    case RTGGX_BUF_BACKBUFFER: *ptr = c->backbuffer; *bytes = n * 4; return 0;
    case RTGGX_BUF_SH_COEFFS: *ptr = c->sh; *bytes = 108; return 0;
    case RTGGX_BUF_BVH_NODES0: case RTGGX_BUF_BVH_NODES1: { const MeshDev& m = c->mesh[id == RTGGX_BUF_BVH_NODES1]; *ptr = m.nodes; *bytes = m.numTris > 1 && m.nodes ? (size_t)(m.numTris - 1) * 64 : 0; return 0; }
    case RTGGX_BUF_BVH_TRIS0: case RTGGX_BUF_BVH_TRIS1: { const MeshDev& m = c->mesh[id == RTGGX_BUF_BVH_TRIS1]; *ptr = m.tris; *bytes = m.tris ? (size_t)m.numTris * 64 : 0; return 0; }
    case RTGGX_BUF_TLAS: *ptr = nullptr; *bytes = 128; return 0;
    case RTGGX_BUF_BVH4_NODES0: case RTGGX_BUF_BVH4_NODES1: { const MeshDev& m = c->mesh[id == RTGGX_BUF_BVH4_NODES1]; *ptr = m.nodes4; *bytes = m.numTris > 1 && m.nodes4 ? (size_t)(m.numTris - 1) * sizeof(Bvh4Node) : 0; return 0; }
    case RTGGX_BUF_BVH4_TOP0: case RTGGX_BUF_BVH4_TOP1: { const MeshDev& m = c->mesh[id == RTGGX_BUF_BVH4_TOP1]; *ptr = m.top; *bytes = m.top ? (size_t)m.topCount * sizeof(Bvh4Node) : 0; return 0; }
    case RTGGX_BUF_BIN_WORK: *ptr = c->binWork; *bytes = (size_t)(((c->W + 15) / 16) * ((c->H + 15) / 16)) * 4u * 4u; return 0;
    case RTGGX_BUF_ENV: *ptr = c->env.texels; *bytes = (size_t)c->env.totalTexels * 8; return 0;
    case RTGGX_BUF_EXCHANGE_TOKENS: *ptr = c->exchangeTokens; *bytes = 4 * 2 * RT_MAX_PEERS; return 0;
    default: setError("unknown buffer id %d", id); return -1;
  }
}

int rtggx_buffer_size(rtggx_context* c, int id, size_t* bytes) { RT_CHECK_CTX(c); void* p; return bufferInfo(c, id, &p, bytes); }

int rtggx_buffer_ptr(rtggx_context* c, int id, void** dptr) {
  RT_CHECK_CTX(c);
  size_t bytes;
  const int r = bufferInfo(c, id, dptr, &bytes);
  if (r) return r;
  if (id == RTGGX_BUF_VISIBILITY || id == RTGGX_BUF_DEPTH) *dptr = c->visDepth;   // packed u64: (depth << 32) | visibility
  if (!*dptr) { setError("buffer %d has no device storage", id); return -1; }
  return 0;
}

int rtggx_readback(rtggx_context* c, int id, void* dst, size_t bytes) {
  RT_CHECK_CTX(c);
  void* p; size_t need;
  int r = bufferInfo(c, id, &p, &need);
  if (r) return r;
  if (bytes < need) { setError("rtggx_readback: buffer %d needs %zu bytes, %zu given", id, need, bytes); return -1; }
  RT_HIP(syncStreams(c));
  if (id == RTGGX_BUF_TLAS) { memcpy(dst, c->invWorld, 128); return 0; }
  if (id == RTGGX_BUF_VISIBILITY || id == RTGGX_BUF_DEPTH) {
    uint32_t *dVis, *dDepth;
    RT_HIP(hipMalloc(&dVis, need)); RT_HIP(hipMalloc(&dDepth, need));
    r = unpackVisDepth(c, dVis, dDepth, c->streamMain);
    if (!r && hipStreamSynchronize(c->streamMain) != hipSuccess) { setError("readback: stream sync failed"); r = -2; }   // the copy below runs on the null stream
    if (!r) { hipError_t e = hipMemcpy(dst, id == RTGGX_BUF_VISIBILITY ? dVis : dDepth, need, hipMemcpyDeviceToHost); if (e != hipSuccess) { setError("hipMemcpy: %s", hipGetErrorString(e)); r = -2; } }
    hipFree(dVis); hipFree(dDepth);
    return r;
  }
  if (need == 0) return 0;
  RT_HIP(hipMemcpy(dst, p, need, hipMemcpyDeviceToHost));
  return 0;
}

int rtggx_upload(rtggx_context* c, int id, const void* src, size_t bytes) {
  RT_CHECK_CTX(c);
  void* p; size_t need;
  int r = bufferInfo(c, id, &p, &need);
  if (r) return r;
  if (bytes != need) { setError("rtggx_upload: buffer %d is %zu bytes, %zu given", id, need, bytes); return -1; }
  RT_HIP(syncStreams(c));
  c->toneMapDone = false;      // (a tone map after an upload reads what was uploaded)
  c->visFlags[c->frameCounter % RT_VIS_RING].rasterFrame = 0u;      // ... and the tiles' words of this frame's visibility pass do not describe it (rtggx_context.h visDirtyBuf)
  if (id == RTGGX_BUF_VISIBILITY || id == RTGGX_BUF_DEPTH) {
    // replace one half of the packed buffer
    uint32_t *dVis, *dDepth;
    RT_HIP(hipMalloc(&dVis, need)); RT_HIP(hipMalloc(&dDepth, need));
    r = unpackVisDepth(c, dVis, dDepth, c->streamMain);
    if (!r && hipStreamSynchronize(c->streamMain) != hipSuccess) { setError("upload: stream sync failed"); r = -2; }
    if (!r) { hipError_t e = hipMemcpy(id == RTGGX_BUF_VISIBILITY ? dVis : dDepth, src, need, hipMemcpyHostToDevice); if (e != hipSuccess) { setError("hipMemcpy: %s", hipGetErrorString(e)); r = -2; } }
    if (!r) r = packVisDepth(c, dVis, dDepth, c->streamMain);
    if (!r && id == RTGGX_BUF_DEPTH && hipMemcpy(c->depth32, src, need, hipMemcpyHostToDevice) != hipSuccess) { setError("upload: depth copy failed"); r = -2; }      // the filters' copy (ray generation writes it otherwise)
    hipStreamSynchronize(c->streamMain);
    hipFree(dVis); hipFree(dDepth);
    return r;
  }
  if (id == RTGGX_BUF_SH_COEFFS) c->shDone = true;
  if (!p || id >= RTGGX_BUF_BVH_NODES0) { setError("rtggx_upload: buffer %d is not writable", id); return -1; }
  RT_HIP(hipMemcpy(p, src, need, hipMemcpyHostToDevice));
  return 0;
}

// Attainable HBM bandwidth of the device, for the roofline's "peak measured beside the vendor figure" (SURVEY 8d): a float4
// copy kernel over two buffers of `bytes` each (far larger than the 256 MiB Infinity Cache when bytes >= 1 GiB: at 2 x 128 MiB the same
// loop reads 7.2 TB/s), timed with events on the main stream; gbytes_per_s = (bytes read + bytes written) / time.  The launch shape
// matters by 20 % on this part and non-temporal accesses by another 8 % (tools/microbench/copy_bw.hip, profiles/r03_h_copy_peak.txt,
// r04_b_copy_peak.txt: a grid-stride loop at 4 workgroups per CU 5.7 TB/s, at 16 per CU 4.6, hipMemcpyAsync 4.8; the same loop at 4 per CU
// with non-temporal loads and stores 6.17 TB/s -- 98 % of the 6.29 MI355X_MICROARCH.md quotes; rounds 2-3 reported 5.5-5.7 without them):
// the shapes of kCopyShapes each get `iterations` launches, plain and non-temporal, and the best one is reported.
static const uint32_t kCopyShapes[] = {2u, 4u, 8u, 16u};   // workgroups per CU
int rtggx_copy_bandwidth(rtggx_context* c, size_t bytes, int iterations, double* gbytesPerS) {
  RT_CHECK_CTX(c);
  if (!gbytesPerS || bytes < 1024 || iterations < 1) { setError("rtggx_copy_bandwidth: bad arguments"); return -1; }
  RT_HIP(syncStreams(c));
  const size_t n = bytes / 16;
  float4 *src = nullptr, *dst = nullptr;
  RT_HIP(hipMalloc(&src, n * 16));
  if (hipMalloc(&dst, n * 16) != hipSuccess) { hipFree(src); setError("rtggx_copy_bandwidth: out of device memory"); return -2; }
  hipMemsetAsync(src, 0x3C, n * 16, c->streamMain);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipError_t e = hipSuccess;
  double best = 0.0;
  for (uint32_t shape : kCopyShapes) {
    const uint32_t blocks = c->numCUs * shape;
    for (int nt = 0; nt < 2 && e == hipSuccess; ++nt) {
      const auto launch = [&]() { if (nt) hipLaunchKernelGGL(copyKernelNT, dim3(blocks), dim3(256), 0, c->streamMain, (const CopyVec4*)src, (CopyVec4*)dst, n);
                                  else hipLaunchKernelGGL(copyKernel, dim3(blocks), dim3(256), 0, c->streamMain, (const float4*)src, dst, n); };
      for (int i = 0; i < 2; ++i) launch();
      hipEventRecord(e0, c->streamMain);
      for (int i = 0; i < iterations; ++i) launch();
      hipEventRecord(e1, c->streamMain);
      e = hipEventSynchronize(e1);
      float ms = 0.0f;
      if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
      if (e != hipSuccess || !(ms > 0.0f)) { if (e == hipSuccess) e = hipErrorUnknown; break; }
      const double rate = 2.0 * (double)(n * 16) * iterations / ((double)ms * 1e-3) / 1e9;
      if (rate > best) best = rate;
    }
    if (e != hipSuccess) break;
  }
  hipEventDestroy(e0); hipEventDestroy(e1); hipFree(src); hipFree(dst);
  if (e != hipSuccess) { setError("rtggx_copy_bandwidth: %s", hipGetErrorString(e)); return -2; }
  *gbytesPerS = best;
  return 0;
}

// Diagnostic: the shader clock the chip is running at right now -- one wave on stream R idles for ~20 us between two readings of
// s_memtime (shader cycles) and s_memrealtime (100 MHz), while whatever the other streams hold keeps running (profiles/r02_*).
__global__ void clockProbeKernel(unsigned long long* out) {
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long r1 = r0;
  while (r1 - r0 < 2000ull) { __builtin_amdgcn_s_sleep(32); r1 = __builtin_amdgcn_s_memrealtime(); }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
}
int rtggx_debug_shader_clock(rtggx_context* c, double* mhz) {
  RT_CHECK_CTX(c);
  if (!mhz) { setError("rtggx_debug_shader_clock: null result"); return -1; }
  unsigned long long* d = nullptr; unsigned long long h[2] = {0, 1};
  RT_HIP(hipMalloc(&d, 16));
  hipLaunchKernelGGL(clockProbeKernel, dim3(1), dim3(64), 0, c->streamRefit, d);
  RT_HIP(hipMemcpyAsync(h, d, 16, hipMemcpyDeviceToHost, c->streamRefit));
  RT_HIP(hipStreamSynchronize(c->streamRefit));
  hipFree(d);
  *mhz = (double)h[0] / (double)h[1] * 100.0;
  return 0;
}

int rtggx_frame_parity(rtggx_context* c, uint32_t* parity) { RT_CHECK_CTX(c); *parity = c->frameParity; return 0; }
int rtggx_bvh_root(rtggx_context* c, uint32_t slot, int32_t* root) { RT_CHECK_CTX(c); if (slot > 1) { setError("bad slot"); return -1; } *root = c->mesh[slot].root; return 0; }

int rtggx_trace_rays(rtggx_context* c, const float* rays, uint32_t n, float* out) {
  RT_CHECK_CTX(c);
  if (!c->asBuilt || !c->haveConstants) { setError("rtggx_trace_rays: build_as / update_frame / update_as first"); return -1; }
  RT_HIP(syncStreams(c));   // the ray bins are shared with the frame path on stream B
  if (c->sceneDirty) { const int r = uploadScene(c, c->streamMain); if (r) return r; }
  { const int r = ensureParams(c); if (r) return r; }
  float *dR, *dO;
  RT_HIP(hipMalloc(&dR, (size_t)n * 32)); RT_HIP(hipMalloc(&dO, (size_t)n * 24));
  RT_HIP(hipMemcpy(dR, rays, (size_t)n * 32, hipMemcpyHostToDevice));
  int r = 0;
  const uint32_t perLaunch = c->numBinsMax * c->binSlots;
  for (uint32_t done = 0; done < n && !r; done += perLaunch) {   // the ray bins' capacity per launch
    const uint32_t m = n - done < perLaunch ? n - done : perLaunch;
    r = launchTraceRays(c, c->slots[c->slot], dR + (size_t)done * 8, m, dO + (size_t)done * 6, c->streamMain);
  }
  if (!r) { hipError_t e = hipStreamSynchronize(c->streamMain); if (e == hipSuccess) e = hipMemcpy(out, dO, (size_t)n * 24, hipMemcpyDeviceToHost); if (e != hipSuccess) { setError("trace_rays: %s", hipGetErrorString(e)); r = -2; } }
  hipFree(dR); hipFree(dO);
  return r;
}

}  // extern "C"
