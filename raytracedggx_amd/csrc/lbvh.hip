// Bottom-level acceleration structures: the gfx950 replacement of the driver-side BLAS build the
// reference requests in RayTracer::buildAccelerationStructures / BuildAccelerationStructures
// (RayTracedGGX/Content/RayTracer.cpp:676-716, 158-233; PREFER_FAST_TRACE, one triangle geometry
// per mesh, R32G32B32_FLOAT positions at stride 24, 32-bit indices) and of its per-frame update (:326-341).
//
// The build, all of it on the device (round 3; rounds 1 and 2 drove the clustering from the host, one synchronisation per round):
//   vertex bounds -> 30-bit Morton codes of the triangle-box centres -> stable LSD radix sort (4 x 8 bits, wave64 ballot ranking)
//   -> PLOC clustering (Meister & Bittner 2018): a fixed number of multi-workgroup rounds, each three kernels that read the number of
//      clusters left from device memory, then ONE workgroup that finishes whatever is left (normally <= 2048 clusters) round by round
//   -> the refit schedule ("treelets"), the list of nodes at the tree's top, the tree's depth and cost
//   -> 64-byte binary nodes (for the oracle / tests), their 4-wide collapse (for the trace kernel), 64-byte leaf triangles in Morton order.
// ~75 kernel launches on one stream for the bunny, no host round trip; the host reads five counts (BuildResult) when the last one has
// ended.  rtggx_build_as issues them all and waits once; a mesh that deforms is REBUILT the same way beside the frames (startRebuild /
// continueRebuild: a few launches per frame on the refit stream, the new topology swapped in between two frames).  What is on the
// per-frame path of such a mesh is refitLbvh: new leaf triangles and boxes for the existing topology, six kernels, no host involvement.
#include <algorithm>
#include <cstring>
#include <functional>
#include <utility>
#include "rtggx_context.h"

namespace rt {

RT_DEV uint32_t expandBits10(uint32_t v) {
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}

// Vertex bounds (the Morton normalisation box) on the device: floats ordered as unsigned integers, one atomic per wave and bound.
RT_DEV uint32_t orderedBits(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
RT_DEV float fromOrderedBits(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }
__global__ void __launch_bounds__(256) boundsKernel(const float* __restrict__ verts, uint32_t nv, uint32_t* __restrict__ bounds /* min[3] max[3], ordered bits */) {
  __shared__ float red[4][6];
  float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
  for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += gridDim.x * blockDim.x)
    for (int k = 0; k < 3; ++k) { const float x = verts[6 * (size_t)v + k]; mn[k] = fminf(mn[k], x); mx[k] = fmaxf(mx[k], x); }
  for (int k = 0; k < 3; ++k) {
    for (int o = 32; o > 0; o >>= 1) { mn[k] = fminf(mn[k], __shfl_down(mn[k], o)); mx[k] = fmaxf(mx[k], __shfl_down(mx[k], o)); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][k] = mn[k]; red[threadIdx.x >> 6][3 + k] = mx[k]; }
  }
  __syncthreads();
  if (threadIdx.x < 3) atomicMin(&bounds[threadIdx.x], orderedBits(fminf(fminf(red[0][threadIdx.x], red[1][threadIdx.x]), fminf(red[2][threadIdx.x], red[3][threadIdx.x]))));
  else if (threadIdx.x < 6) atomicMax(&bounds[threadIdx.x], orderedBits(fmaxf(fmaxf(red[0][threadIdx.x], red[1][threadIdx.x]), fmaxf(red[2][threadIdx.x], red[3][threadIdx.x]))));
}
__global__ void mortonKernel(const float* __restrict__ verts, const uint32_t* __restrict__ idx, uint32_t n,
                             const uint32_t* __restrict__ bounds, uint32_t* __restrict__ codes, uint32_t* __restrict__ order,
                             float* __restrict__ triBox) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float3 bmin, invExt;
  { const float x0 = fromOrderedBits(bounds[0]), y0 = fromOrderedBits(bounds[1]), z0 = fromOrderedBits(bounds[2]);
    const float x1 = fromOrderedBits(bounds[3]), y1 = fromOrderedBits(bounds[4]), z1 = fromOrderedBits(bounds[5]);
    bmin = make_float3(x0, y0, z0);
    invExt = make_float3(x1 > x0 ? 1.0f / (x1 - x0) : 0.0f, y1 > y0 ? 1.0f / (y1 - y0) : 0.0f, z1 > z0 ? 1.0f / (z1 - z0) : 0.0f); }
  float mn[3], mx[3];
  for (int k = 0; k < 3; ++k) {
    const float a = verts[6 * (size_t)idx[3 * i] + k], b = verts[6 * (size_t)idx[3 * i + 1] + k], c = verts[6 * (size_t)idx[3 * i + 2] + k];
    mn[k] = fminf(a, fminf(b, c)); mx[k] = fmaxf(a, fmaxf(b, c));
    triBox[6 * (size_t)i + k] = mn[k]; triBox[6 * (size_t)i + 3 + k] = mx[k];
  }
  const float cx = (0.5f * (mn[0] + mx[0]) - bmin.x) * invExt.x;
  const float cy = (0.5f * (mn[1] + mx[1]) - bmin.y) * invExt.y;
  const float cz = (0.5f * (mn[2] + mx[2]) - bmin.z) * invExt.z;
  const uint32_t x = (uint32_t)fminf(fmaxf(cx * 1024.0f, 0.0f), 1023.0f);
  const uint32_t y = (uint32_t)fminf(fmaxf(cy * 1024.0f, 0.0f), 1023.0f);
  const uint32_t z = (uint32_t)fminf(fmaxf(cz * 1024.0f, 0.0f), 1023.0f);
  codes[i] = (expandBits10(x) << 2) | (expandBits10(y) << 1) | expandBits10(z);
  order[i] = i;
}

// ---- stable LSD radix sort, 8 bits per pass, 256 keys per workgroup --------------------------------
__global__ void __launch_bounds__(256) radixHist(const uint32_t* __restrict__ keys, uint32_t n, int shift, uint32_t* __restrict__ hist, uint32_t numBlocks) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u);
  __syncthreads();
  hist[threadIdx.x * numBlocks + blockIdx.x] = h[threadIdx.x];
}
// Exclusive scan of `count` values in two steps (a single workgroup took 106 us per pass for the bunny's 70 000 histogram entries,
// a third of the whole build): scanChunks -- one workgroup per 1024 values: the exclusive scan inside the chunk and the chunk's total --,
// scanTotals -- one workgroup: the exclusive scan of the totals.  A consumer adds chunkSums[index >> 10] to the value it reads.
__global__ void __launch_bounds__(1024) scanChunks(uint32_t* __restrict__ data, uint32_t count, uint32_t* __restrict__ chunkSums) {
  __shared__ uint32_t waveTotal[16];
  const uint32_t i = blockIdx.x * 1024u + threadIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t v = i < count ? data[i] : 0u;
  uint32_t inc = v;
  for (int o = 1; o < 64; o <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)inc, o); if ((int)lane >= o) inc += u; }
  if (lane == 63u) waveTotal[wave] = inc;
  __syncthreads();
  uint32_t base = 0;
  for (uint32_t w = 0; w < wave; ++w) base += waveTotal[w];
  if (i < count) data[i] = base + inc - v;
  if (threadIdx.x == 1023u) chunkSums[blockIdx.x] = base + inc;
}
__global__ void __launch_bounds__(1024) scanTotals(uint32_t* __restrict__ sums, uint32_t n) {
  __shared__ uint32_t waveTotal[16];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t carry = 0;
  for (uint32_t c0 = 0; c0 < n; c0 += 1024u) {
    const uint32_t i = c0 + threadIdx.x;
    const uint32_t v = i < n ? sums[i] : 0u;
    uint32_t inc = v;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)inc, o); if ((int)lane >= o) inc += u; }
    if (lane == 63u) waveTotal[wave] = inc;
    __syncthreads();
    uint32_t base = carry, total = 0;
    for (uint32_t w = 0; w < 16u; ++w) { if (w < wave) base += waveTotal[w]; total += waveTotal[w]; }
    if (i < n) sums[i] = base + inc - v;
    carry += total;
    __syncthreads();
  }
}
__global__ void __launch_bounds__(256) radixScatter(const uint32_t* __restrict__ keysIn, const uint32_t* __restrict__ valsIn, uint32_t n, int shift,
                                                    const uint32_t* __restrict__ hist, const uint32_t* __restrict__ chunkSums, uint32_t numBlocks,
                                                    uint32_t* __restrict__ keysOut, uint32_t* __restrict__ valsOut) {
  __shared__ uint32_t waveCount[4][256];
  for (int w = 0; w < 4; ++w) waveCount[w][threadIdx.x] = 0;
  __syncthreads();
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  const bool active = i < n;
  const uint32_t key = active ? keysIn[i] : 0xFFFFFFFFu;
  const uint32_t digit = (key >> shift) & 255u;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  unsigned long long peers = __ballot(active);
  for (int b = 0; b < 8; ++b) {
    const unsigned long long m = __ballot(active && ((digit >> b) & 1u));
    peers &= ((digit >> b) & 1u) ? m : ~m;
  }
  const uint32_t rankInWave = __popcll(peers & ((1ull << lane) - 1ull));
  if (active && rankInWave == 0) waveCount[wave][digit] = (uint32_t)__popcll(peers);
  __syncthreads();
  if (active) {
    const uint32_t h = digit * numBlocks + blockIdx.x;
    uint32_t off = hist[h] + chunkSums[h >> 10] + rankInWave;
    for (uint32_t w = 0; w < wave; ++w) off += waveCount[w][digit];
    keysOut[off] = key; valsOut[off] = valsIn[i];
  }
}

// ---- PLOC: parallel locally-ordered clustering (Meister & Bittner 2018) ------------------------------------
// Agglomerative build over the Morton-ordered triangles: every cluster looks RT_PLOC_RADIUS positions to either side
// for the partner that gives the smallest merged box; mutual choices merge into a new node; the survivors are
// compacted (order kept) and the step repeats until one cluster is left.  Trees come out close to a SAH sweep
// build -- on the bunny a third fewer traversal steps per ray than the Karras radix tree (profiles/r01_d).
// Everything is deterministic: ties go to the lower position, node indices come from prefix sums.
//
// Device-driven (round 3).  How many clusters a round leaves is known on the device only, so nothing on the host depends on it:
// the host issues a FIXED number of rounds sized from the triangle count (each round: plocNearest, plocScatter, both with
// the grid of the first round -- workgroups beyond the clusters left leave at once), then plocFinal, ONE workgroup that runs rounds
// until a single cluster is left, whatever it is handed (normally <= 2048 clusters; if the fixed rounds merged less than expected it
// simply has more to do).  The state between rounds -- clusters left, next node index, rounds so far -- lives in a two-entry ring in
// device memory (round r reads entry r & 1 and writes the other); roundBase[k] = first node of round k, for the refit schedule.
#define RT_PLOC_RADIUS 16
#define RT_MAX_ROUNDS 1024
#define RT_TREELET_NODES 1024
// plocFinal's LDS holds RT_PLOC_LDS clusters (44 bytes each: 135 KB of the 160 KB a gfx950 workgroup may have).  The multi-workgroup rounds
// before it stop by themselves once RT_PLOC_STOP clusters are left: the host issues rounds as if every round kept RT_PLOC_KEEP of its
// clusters, and a round that finds the list short enough returns at once and hands the state on.  Measured (profiles/r03_g_build.txt):
// a round in LDS costs ~6 us at 3000 clusters and ~2 at 1000, a multi-workgroup round ~13 us of kernels + two launch gaps, a round of ONE
// workgroup on the lists in global memory ~90 us -- so the list should reach plocFinal below its LDS capacity, but not far below.
#define RT_PLOC_LDS 3072
#define RT_PLOC_STOP 2048
#define RT_PLOC_KEEP 0.80
struct PlocState { uint32_t m, nodeBase, rounds, buf; };      // clusters left, next node index, rounds done, which of the two cluster lists holds them
struct PlocArrays {      // what a round reads and writes (by value to the kernels)
  int32_t* clRef[2]; float* clBox[2]; int32_t* nn;
  int32_t *left, *right, *nodeParent, *leafParent; float* nodeBox; uint32_t* cnt[RT_TREELET_LEVELS];
  uint32_t* roundBase; BuildResult* res;
};
RT_DEV float mergedArea(const float* a, const float* b) {
  const float ex = fmaxf(a[3], b[3]) - fminf(a[0], b[0]), ey = fmaxf(a[4], b[4]) - fminf(a[1], b[1]), ez = fmaxf(a[5], b[5]) - fminf(a[2], b[2]);
  return (ex * ey + ey * ez) + ez * ex;
}
// first kernel of a build: the device records start clean
__global__ void buildBegin(uint32_t n, PlocState* state, BuildResult* res, uint32_t* bounds) {
  if (threadIdx.x == 0) {
    state[0].m = n; state[0].nodeBase = 0u; state[0].rounds = 0u; state[0].buf = 0u; state[1] = state[0];
    BuildResult z{}; *res = z;
    for (int k = 0; k < 3; ++k) { bounds[k] = 0xFFFFFFFFu; bounds[3 + k] = 0u; }
  }
}
__global__ void plocInit(int n, const uint32_t* __restrict__ order, const float* __restrict__ triBox, int32_t* __restrict__ clRef, float* __restrict__ clBox) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  clRef[s] = ~s;
  for (int k = 0; k < 6; ++k) clBox[6 * (size_t)s + k] = triBox[6 * (size_t)order[s] + k];
}
RT_DEV int plocNearestOf(int i, int m, int radius, const float* __restrict__ clBox) {
  float mine[6];
  for (int k = 0; k < 6; ++k) mine[k] = clBox[6 * (size_t)i + k];
  float best = __builtin_inff(); int bj = -1;
  const int lo = max(i - radius, 0), hi = min(i + radius, m - 1);
  for (int j = lo; j <= hi; ++j) {
    if (j == i) continue;
    const float d = mergedArea(mine, clBox + 6 * (size_t)j);
    if (d < best) { best = d; bj = j; }
  }
  return bj;
}
// A round's first kernel: nearest neighbours, the keep / merge flags they imply, and the workgroup's counts of both.  A workgroup
// owns 256 positions and stages the boxes of those and of 2 x radius positions to either side in LDS: the flags of position i need the
// nearest neighbour of i's nearest neighbour (within the radius of i), whose own search reaches another radius further.  (Round 3
// began with two kernels -- nearest, then flags + counts --: with ~14 rounds to issue, a launch less per round is 100 us of the build.)
#define RT_PLOC_APRON (2 * RT_PLOC_RADIUS)
#define RT_PLOC_LANES 4      // lanes that share one position's search (each a quarter of the neighbourhood): there are only ~1000 waves of work in
                             // the largest round, one per SIMD, and a wave's search is a chain of dependent LDS reads -- four times the waves, each a quarter as long
// The nearest neighbour of position i among [i - radius, i + radius] (boxes in LDS, entry e = position lo + e): smallest merged box, ties
// to the lower position.  RT_PLOC_LANES consecutive lanes call it together for the same i, `sub` = the caller's share.
RT_DEV int plocNearestShared(const float (*box)[256 + 2 * RT_PLOC_APRON], int i, int lo, int m, int radius, uint32_t sub) {
  float best = __builtin_inff(); int bj = 0x7FFFFFFF;
  if (i >= 0 && i < m) {
    const int ei = i - lo;
    float mine[6];
    for (int k = 0; k < 6; ++k) mine[k] = box[k][ei];
    const int jl = max(i - radius, 0), jh = min(i + radius, m - 1);
    for (int j = jl + (int)sub; j <= jh; j += RT_PLOC_LANES) {
      if (j == i) continue;
      const int ej = j - lo;
      const float ex = fmaxf(mine[3], box[3][ej]) - fminf(mine[0], box[0][ej]), ey = fmaxf(mine[4], box[4][ej]) - fminf(mine[1], box[1][ej]), ez = fmaxf(mine[5], box[5][ej]) - fminf(mine[2], box[2][ej]);
      const float d = (ex * ey + ey * ez) + ez * ex;
      if (d < best) { best = d; bj = j; }      // (ascending j: the first of equal distances stays)
    }
  }
  for (int o = 1; o < RT_PLOC_LANES; o <<= 1) {      // the best of the group's shares: smaller distance, then lower position
    const float od = __shfl_xor(best, o); const int oj = __shfl_xor(bj, o);
    if (od < best || (od == best && oj < bj)) { best = od; bj = oj; }
  }
  return bj == 0x7FFFFFFF ? -1 : bj;
}
__global__ void __launch_bounds__(256 * RT_PLOC_LANES) plocNearest(const PlocState* __restrict__ state, uint32_t r, int radius, PlocArrays A, uint2* __restrict__ blockCounts) {
  __shared__ float box[6][256 + 2 * RT_PLOC_APRON];
  __shared__ int32_t nnS[256 + 2 * RT_PLOC_RADIUS];
  __shared__ uint32_t wk[4], wm[4];
  const PlocState st = state[r & 1u];
  const int m = (int)st.m;
  if (m <= RT_PLOC_STOP || (int)(blockIdx.x * 256) >= m) return;      // (a list this short is left to plocFinal: this round is skipped)
  if (radius > RT_PLOC_RADIUS) radius = RT_PLOC_RADIUS;
  const float* __restrict__ clBox = A.clBox[st.buf];
  const int b0 = (int)(blockIdx.x * 256), lo = b0 - RT_PLOC_APRON;      // LDS entry e holds position lo + e
  for (int e = threadIdx.x; e < 256 + 2 * RT_PLOC_APRON; e += 256 * RT_PLOC_LANES) {
    const int pos = lo + e;
    if (pos >= 0 && pos < m) for (int k = 0; k < 6; ++k) box[k][e] = clBox[6 * (size_t)pos + k];
  }
  __syncthreads();
  // nearest neighbour of positions [b0 - radius, b0 + 256 + radius)
  for (int q = threadIdx.x / RT_PLOC_LANES; q < 256 + 2 * RT_PLOC_RADIUS; q += 256) {      // (uniform per group of lanes; whole waves run the same number of rounds)
    const int bj = plocNearestShared(box, b0 - RT_PLOC_RADIUS + q, lo, m, radius, threadIdx.x % RT_PLOC_LANES);
    if (threadIdx.x % RT_PLOC_LANES == 0) nnS[q] = bj;
  }
  __syncthreads();
  const int i = b0 + (int)threadIdx.x;
  uint32_t keep = 0u, merge = 0u;
  if (threadIdx.x < 256 && i < m) {
    const int j = nnS[i - b0 + RT_PLOC_RADIUS];
    A.nn[i] = j;
    const bool mutual = j >= 0 && nnS[j - b0 + RT_PLOC_RADIUS] == i;
    merge = mutual && i < j ? 1u : 0u;
    keep = mutual && i > j ? 0u : 1u;
  }
  const unsigned long long bk = __ballot(keep != 0u), bm = __ballot(merge != 0u);
  if (threadIdx.x < 256 && (threadIdx.x & 63u) == 0u) { wk[threadIdx.x >> 6] = (uint32_t)__popcll(bk); wm[threadIdx.x >> 6] = (uint32_t)__popcll(bm); }
  __syncthreads();
  if (threadIdx.x == 0) blockCounts[blockIdx.x] = make_uint2(wk[0] + wk[1] + wk[2] + wk[3], wm[0] + wm[1] + wm[2] + wm[3]);
}
// keep / merge flags of position i: mutual nearest neighbours merge, the lower position carries the new node, the higher one disappears
RT_DEV void plocFlags(int i, int m, const int32_t* __restrict__ nn, int& j, bool& mutual, uint32_t& keep, uint32_t& merge) {
  j = -1; mutual = false; keep = 0u; merge = 0u;
  if (i >= m) return;
  j = nn[i];
  mutual = j >= 0 && nn[j] == i;
  merge = mutual && i < j ? 1u : 0u;
  keep = mutual && i > j ? 0u : 1u;
}
// Position i survives at position p of the next round's list: as itself, or merged with j into node `node`.  A new node also gets, per
// treelet level, the number of nodes of its subtree that level still has to place (its children are older: their counts are final).
RT_DEV void plocEmit(int i, int j, bool mutual, uint32_t p, int node, const int32_t* __restrict__ clRef, const float* __restrict__ clBox,
                     int32_t* __restrict__ clRefOut, float* __restrict__ clBoxOut, const PlocArrays& A) {
  if (mutual) {
    const int32_t L = clRef[i], R = clRef[j];
    A.left[node] = L; A.right[node] = R;
    if (L < 0) A.leafParent[~L] = node; else A.nodeParent[L] = node;
    if (R < 0) A.leafParent[~R] = node; else A.nodeParent[R] = node;
    for (int k = 0; k < 3; ++k) {
      const float mn = fminf(clBox[6 * (size_t)i + k], clBox[6 * (size_t)j + k]), mx = fmaxf(clBox[6 * (size_t)i + 3 + k], clBox[6 * (size_t)j + 3 + k]);
      A.nodeBox[6 * (size_t)node + k] = mn; A.nodeBox[6 * (size_t)node + 3 + k] = mx;
      clBoxOut[6 * (size_t)p + k] = mn; clBoxOut[6 * (size_t)p + 3 + k] = mx;
    }
    clRefOut[p] = node;
    uint32_t below = 0xFFFFFFFFu;      // level 0 places every node
    for (int l = 0; l < RT_TREELET_LEVELS; ++l) {
      const uint32_t c = below > RT_TREELET_NODES ? 1u + (L >= 0 ? A.cnt[l][L] : 0u) + (R >= 0 ? A.cnt[l][R] : 0u) : 0u;
      A.cnt[l][node] = c; below = c;
    }
  } else {
    clRefOut[p] = clRef[i];
    for (int k = 0; k < 6; ++k) clBoxOut[6 * (size_t)p + k] = clBox[6 * (size_t)i + k];
  }
}
__global__ void __launch_bounds__(256) plocScatter(PlocState* __restrict__ state, uint32_t r, const uint2* __restrict__ blockCounts, PlocArrays A) {
  const PlocState st = state[r & 1u];
  const int m = (int)st.m;
  if (m <= RT_PLOC_STOP) { if (blockIdx.x == 0 && threadIdx.x == 0) state[(r + 1u) & 1u] = st; return; }      // skipped (plocNearest): the state is handed on as it is
  const uint32_t nblocks = ((uint32_t)m + 255u) / 256u;
  if (blockIdx.x >= nblocks) return;
  // kept positions / merges in the workgroups before this one
  __shared__ uint32_t redK[4], redM[4], wk[4], wm[4];
  uint32_t ok = 0, om = 0;
  for (uint32_t b = threadIdx.x; b < blockIdx.x; b += 256u) { const uint2 c = blockCounts[b]; ok += c.x; om += c.y; }
  for (int o = 32; o > 0; o >>= 1) { ok += __shfl_down(ok, o); om += __shfl_down(om, o); }
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 256 + threadIdx.x;
  int j; bool mutual; uint32_t keep, merge;
  plocFlags(i, m, A.nn, j, mutual, keep, merge);
  const unsigned long long bk = __ballot(keep != 0u), bm = __ballot(merge != 0u), below = (1ull << lane) - 1ull;
  if (lane == 0u) { redK[wave] = ok; redM[wave] = om; wk[wave] = (uint32_t)__popcll(bk); wm[wave] = (uint32_t)__popcll(bm); }
  __syncthreads();
  const uint32_t baseK = redK[0] + redK[1] + redK[2] + redK[3], baseM = redM[0] + redM[1] + redM[2] + redM[3];
  uint32_t rk = (uint32_t)__popcll(bk & below), rm = (uint32_t)__popcll(bm & below);
  for (uint32_t w = 0; w < wave; ++w) { rk += wk[w]; rm += wm[w]; }
  if (keep) plocEmit(i, j, mutual, baseK + rk, (int)(st.nodeBase + baseM + rm), A.clRef[st.buf], A.clBox[st.buf], A.clRef[st.buf ^ 1u], A.clBox[st.buf ^ 1u], A);
  if (blockIdx.x == nblocks - 1u && threadIdx.x == 0) {      // the last workgroup knows the totals
    PlocState nx;
    nx.m = baseK + wk[0] + wk[1] + wk[2] + wk[3]; nx.nodeBase = st.nodeBase + baseM + wm[0] + wm[1] + wm[2] + wm[3]; nx.rounds = st.rounds + 1u; nx.buf = st.buf ^ 1u;
    if (st.rounds < RT_MAX_ROUNDS) A.roundBase[st.rounds] = st.nodeBase; else atomicOr(&A.res->error, 1u);
    state[(r + 1u) & 1u] = nx;
  }
}
// One workgroup finishes the clustering: rounds until one cluster is left (each merges at least one pair: it ends), then the root's
// parent, the last roundBase entry and the round count.  `r`: the rounds issued before it (it reads state entry r & 1).
// While more than RT_PLOC_LDS clusters are left the rounds run as above, on the lists in global memory; from then on the clusters --
// reference, box, subtree counts -- live in LDS: the tail of the clustering is ~40 rounds that merge a handful of pairs each, and
// from global memory a round of ONE workgroup is a chain of L2 round trips (9.5 us per round, 360 us for the bunny: a quarter of its
// build; in LDS 1 us).  A round compacts the list in place: every thread first reads what it needs into registers, then writes.
struct PlocLds {
  int32_t ref[RT_PLOC_LDS]; float box[6][RT_PLOC_LDS]; uint32_t cnt[RT_TREELET_LEVELS][RT_PLOC_LDS]; int32_t nn[RT_PLOC_LDS];
  uint32_t wk[16], wm[16];
};
__global__ void __launch_bounds__(1024) plocFinal(const PlocState* __restrict__ state, uint32_t r, int radius, uint32_t n, PlocArrays A) {
  __shared__ PlocLds L;
  const PlocState st = state[r & 1u];
  uint32_t m = st.m, nodeBase = st.nodeBase, rounds = st.rounds, p = st.buf;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const unsigned long long below = (1ull << lane) - 1ull;
  while (m > RT_PLOC_LDS) {      // (only when the rounds before merged less than expected)
    const int32_t* clRef = A.clRef[p]; const float* clBox = A.clBox[p];
    for (uint32_t i = threadIdx.x; i < m; i += 1024u) A.nn[i] = plocNearestOf((int)i, (int)m, radius, clBox);
    __syncthreads();
    uint32_t keptBase = 0, mergedBase = 0;
    for (uint32_t c0 = 0; c0 < m; c0 += 1024u) {
      const int i = (int)(c0 + threadIdx.x);
      int j; bool mutual; uint32_t keep, merge;
      plocFlags(i, (int)m, A.nn, j, mutual, keep, merge);
      const unsigned long long bk = __ballot(keep != 0u), bm = __ballot(merge != 0u);
      if (lane == 0u) { L.wk[wave] = (uint32_t)__popcll(bk); L.wm[wave] = (uint32_t)__popcll(bm); }
      __syncthreads();
      uint32_t rk = (uint32_t)__popcll(bk & below), rm = (uint32_t)__popcll(bm & below), totK = 0, totM = 0;
      for (uint32_t w = 0; w < 16u; ++w) { if (w < wave) { rk += L.wk[w]; rm += L.wm[w]; } totK += L.wk[w]; totM += L.wm[w]; }
      if (keep) plocEmit(i, j, mutual, keptBase + rk, (int)(nodeBase + mergedBase + rm), clRef, clBox, A.clRef[p ^ 1u], A.clBox[p ^ 1u], A);
      keptBase += totK; mergedBase += totM;
      __syncthreads();
    }
    if (threadIdx.x == 0) { if (rounds < RT_MAX_ROUNDS) A.roundBase[rounds] = nodeBase; else atomicOr(&A.res->error, 1u); }
    ++rounds; nodeBase += mergedBase; m = keptBase; p ^= 1u;
    __syncthreads();
  }
  if (threadIdx.x == 0) { A.res->finalEntry = st.m; A.res->finalLds = m; }      // (statistics: clusters handed over, clusters at the start of the LDS rounds)
  // the clusters into LDS
  for (uint32_t i = threadIdx.x; i < m; i += 1024u) {
    const int32_t ref = A.clRef[p][i];
    L.ref[i] = ref;
    for (int k = 0; k < 6; ++k) L.box[k][i] = A.clBox[p][6 * (size_t)i + k];
    for (int l = 0; l < RT_TREELET_LEVELS; ++l) L.cnt[l][i] = ref >= 0 ? A.cnt[l][ref] : 0u;
  }
  __syncthreads();
  while (m > 1u) {
    // nearest neighbour within the radius: smallest merged box, ties to the lower position.  As the list shrinks, more lanes share one
    // position's search (`group` consecutive lanes, each every group-th neighbour, then the best of the group): the rounds of the tail
    // are chains of dependent LDS reads, and a thousand threads are there anyway
    uint32_t group = 1u;
    while (group < 32u && group * 2u * m <= 1024u) group *= 2u;
    for (uint32_t i = threadIdx.x / group; i < m; i += 1024u / group) {      // (uniform per wave: m and group are)
      float mine[6];
      for (int k = 0; k < 6; ++k) mine[k] = L.box[k][i];
      float best = __builtin_inff(); int bj = 0x7FFFFFFF;
      const int lo = max((int)i - radius, 0), hi = min((int)i + radius, (int)m - 1);
      for (int j = lo + (int)(threadIdx.x % group); j <= hi; j += (int)group) {
        if (j == (int)i) continue;
        const float ex = fmaxf(mine[3], L.box[3][j]) - fminf(mine[0], L.box[0][j]), ey = fmaxf(mine[4], L.box[4][j]) - fminf(mine[1], L.box[1][j]), ez = fmaxf(mine[5], L.box[5][j]) - fminf(mine[2], L.box[2][j]);
        const float d = (ex * ey + ey * ez) + ez * ex;
        if (d < best) { best = d; bj = j; }
      }
      for (uint32_t o = 1u; o < group; o <<= 1) {
        const float od = __shfl_xor(best, (int)o); const int oj = __shfl_xor(bj, (int)o);
        if (od < best || (od == best && oj < bj)) { best = od; bj = oj; }
      }
      if (threadIdx.x % group == 0u) L.nn[i] = bj == 0x7FFFFFFF ? -1 : bj;
    }
    __syncthreads();
    // flags, ranks and everything a surviving position needs, into registers (two positions per thread) ...
    struct Out { bool keep, mutual; uint32_t pos; int32_t node, l, r; float box[6]; uint32_t cnt[RT_TREELET_LEVELS]; } out[RT_PLOC_LDS / 1024];
    uint32_t keptBase = 0, mergedBase = 0;
#pragma unroll
    for (uint32_t c = 0; c < RT_PLOC_LDS / 1024; ++c) {      // (unrolled: `out` must stay in registers)
      if (c * 1024u >= m) { out[c].keep = false; continue; }   // uniform
      const int i = (int)(c * 1024u + threadIdx.x);
      int j; bool mutual; uint32_t keep, merge;
      plocFlags(i, (int)m, L.nn, j, mutual, keep, merge);
      const unsigned long long bk = __ballot(keep != 0u), bm = __ballot(merge != 0u);
      if (lane == 0u) { L.wk[wave] = (uint32_t)__popcll(bk); L.wm[wave] = (uint32_t)__popcll(bm); }
      __syncthreads();
      uint32_t rk = (uint32_t)__popcll(bk & below), rm = (uint32_t)__popcll(bm & below), totK = 0, totM = 0;
      for (uint32_t w = 0; w < 16u; ++w) { if (w < wave) { rk += L.wk[w]; rm += L.wm[w]; } totK += L.wk[w]; totM += L.wm[w]; }
      Out& o = out[c];
      o.keep = keep != 0u; o.mutual = mutual;
      if (o.keep) {
        o.pos = keptBase + rk; o.node = (int32_t)(nodeBase + mergedBase + rm);
        o.l = L.ref[i];
        if (mutual) {
          o.r = L.ref[j];
          for (int k = 0; k < 3; ++k) { o.box[k] = fminf(L.box[k][i], L.box[k][j]); o.box[3 + k] = fmaxf(L.box[3 + k][i], L.box[3 + k][j]); }
          uint32_t belowCnt = 0xFFFFFFFFu;
          for (int l = 0; l < RT_TREELET_LEVELS; ++l) { const uint32_t cc = belowCnt > RT_TREELET_NODES ? 1u + L.cnt[l][i] + L.cnt[l][j] : 0u; o.cnt[l] = cc; belowCnt = cc; }
        } else {
          for (int k = 0; k < 6; ++k) o.box[k] = L.box[k][i];
          for (int l = 0; l < RT_TREELET_LEVELS; ++l) o.cnt[l] = L.cnt[l][i];
        }
      }
      keptBase += totK; mergedBase += totM;
      __syncthreads();
    }
    // ... then the compacted list in place, and the new nodes
#pragma unroll
    for (uint32_t c = 0; c < RT_PLOC_LDS / 1024; ++c) {
      const Out& o = out[c];
      if (!o.keep) continue;
      L.ref[o.pos] = o.mutual ? o.node : o.l;
      for (int k = 0; k < 6; ++k) L.box[k][o.pos] = o.box[k];
      for (int l = 0; l < RT_TREELET_LEVELS; ++l) L.cnt[l][o.pos] = o.cnt[l];
      if (o.mutual) {
        A.left[o.node] = o.l; A.right[o.node] = o.r;
        if (o.l < 0) A.leafParent[~o.l] = o.node; else A.nodeParent[o.l] = o.node;
        if (o.r < 0) A.leafParent[~o.r] = o.node; else A.nodeParent[o.r] = o.node;
        for (int k = 0; k < 6; ++k) A.nodeBox[6 * (size_t)o.node + k] = o.box[k];
        for (int l = 0; l < RT_TREELET_LEVELS; ++l) A.cnt[l][o.node] = o.cnt[l];
      }
    }
    if (threadIdx.x == 0) { if (rounds < RT_MAX_ROUNDS) A.roundBase[rounds] = nodeBase; else atomicOr(&A.res->error, 1u); }
    ++rounds; nodeBase += mergedBase; m = keptBase;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    A.roundBase[rounds < RT_MAX_ROUNDS ? rounds : RT_MAX_ROUNDS] = nodeBase;
    A.res->numRounds = rounds < RT_MAX_ROUNDS ? rounds : RT_MAX_ROUNDS;
    if (n > 1u) A.nodeParent[n - 2u] = -1;      // the last node created is the root
  }
}

// Leaf depth = number of ancestors = the most stack entries a traversal reaching that leaf can hold.
__global__ void depthKernel(int n, const int32_t* __restrict__ nodeParent, const int32_t* __restrict__ leafParent, uint32_t* __restrict__ maxDepth) {
  const int leaf = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t d = 0;
  if (leaf < n) for (int cur = leafParent[leaf]; cur >= 0; cur = nodeParent[cur]) ++d;
  for (int o = 32; o > 0; o >>= 1) d = max(d, (uint32_t)__shfl_down((int)d, o));
  __shared__ uint32_t red[16];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) { for (uint32_t w = 1; w < (blockDim.x + 63u) / 64u; ++w) d = max(d, red[w]); if (d) atomicMax(maxDepth, d); }
}

__global__ void emitNodes(int n, const uint32_t* __restrict__ order, const float* __restrict__ triBox, const int32_t* __restrict__ left,
                          const int32_t* __restrict__ right, const float* __restrict__ nodeBox, BvhNode* __restrict__ nodes) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1) return;
  BvhNode nd;
  const int32_t l = left[i], r = right[i];
  const float* lb = l < 0 ? &triBox[6 * (size_t)order[~l]] : &nodeBox[6 * (size_t)l];
  const float* rb = r < 0 ? &triBox[6 * (size_t)order[~r]] : &nodeBox[6 * (size_t)r];
  for (int k = 0; k < 3; ++k) { nd.lmin[k] = lb[k]; nd.lmax[k] = lb[3 + k]; nd.rmin[k] = rb[k]; nd.rmax[k] = rb[3 + k]; }
  nd.left = l; nd.right = r; nd.pad[0] = 0; nd.pad[1] = 0;
  nodes[i] = nd;
}
// (a treelet of the refit schedule, see "the refit schedule" below: the collapse's dynamic programme runs in the same order)
struct RefitTreelet { uint32_t itemBegin, roundBegin, numRounds, pad; };
RT_DEV uint32_t treeletBase(const BuildResult* res, int level) { uint32_t b = 0; for (int l = 0; l < level; ++l) b += res->treelets[l]; return b; }
// ---- the 4-wide collapse (round 4: by surface area) --------------------------------------------------------------------------------
// The reference asks the driver for PREFER_FAST_TRACE (RayTracer.cpp:676-716); what this builder can do for the traversal's count of
// dependent steps is choose WHICH binary nodes become 4-wide nodes.  Rounds 1-3 collapsed by depth parity -- every binary node of even
// depth took its grandchildren: 3.0 entries per node on the bunny; opening the largest child first from the top down gives the same
// (measured: 34 462 nodes against 34 764) -- the nodes are counted at the bottom of the tree, and a top-down rule leaves whatever
// subtrees it ends on.  So the choice is made from the bottom up, by dynamic programming over the binary tree (the idea of Ylitie,
// Karras, Laine 2017, section 3, for width 4): the expected number of node steps of a random ray is proportional to the summed surface
// area of the 4-wide nodes, and
//     F(n, i) = the least such sum with which the subtree of n can be covered by AT MOST i entries (i = 1, 2, 3)
//     F(leaf, i) = 0
//     F(n, 1) = A(n) + min over k = 1..3 of F(left, k) + F(right, 4 - k)          (n is a 4-wide node: its entries split k : 4 - k)
//     F(n, i) = min(F(n, 1), min over k = 1..i-1 of F(left, k) + F(right, i - k))  (n stays one entry, or is opened)
// with the choices recorded beside the sums.  Boxes are still the binary tree's own, so the triangles a ray reaches can only grow
// relative to the binary traversal; a node step costs what it did (seven 16-byte loads), there are fewer of them.
//   collapseCostTreelets   F and the choices for every node, children before parents: the refit schedule's order (one workgroup per
//                          treelet, round by round with the treelet's sums in LDS, level after level)
//   entries4Kernel         every binary node i: the entries E(i) it has IF it is a 4-wide node, by following the choices down
//   roots4Kernel           which binary nodes ARE 4-wide nodes: the root, and every internal entry of one.  Thread i walks its ancestors from
//                          the root down through the E() of the 4-wide nodes on the way: either it arrives at i, or i lies inside one of them.
//                          On the way it adds up what a traversal's stack can hold there (entries - 1 per 4-wide ancestor): BuildResult::stack4.
// All of it is part of the topology (BvhTopo::ent4, lvl4): a refit emits the same nodes with new boxes (emitNodes4).
RT_DEV float halfAreaOf(const float* __restrict__ b) { const float ex = b[3] - b[0], ey = b[4] - b[1], ez = b[5] - b[2]; return (ex * ey + ey * ez) + ez * ex; }
// choices word: bits 0-1 k of F(n, 1); bit 2: F(n, 2) opens n (1 : 1); bits 4-5: F(n, 3) keeps n (0) or opens it k : 3 - k (k = 1, 2)
__global__ void __launch_bounds__(256) collapseCostTreelets(int level, const RefitTreelet* __restrict__ treelets, const int4* __restrict__ items, const uint32_t* __restrict__ roundOfs,
                                                            const float* __restrict__ nodeBox, const uint32_t* __restrict__ cnt0, int32_t root, float areaWeight, float trisWeight,
                                                            float4* __restrict__ cost4, const BuildResult* __restrict__ res) {
  __shared__ float sF[RT_TREELET_NODES][3];
  const uint32_t base = treeletBase(res, level), K = res->treelets[level];
  // a node's cost: areaWeight x A(n) / A(root) + trisWeight x triangles(n) / triangles(root) -- the chance that a random ray enters it, and
  // the chance that a ray STARTING on the surface starts inside it (rtggx_debug_collapse_weights)
  const float ka = areaWeight / halfAreaOf(nodeBox + 6 * (size_t)root), kt = trisWeight / (float)(cnt0[root] + 1u);
  for (uint32_t tk = blockIdx.x; tk < K; tk += gridDim.x) {
    const RefitTreelet tl = treelets[base + tk];
    for (uint32_t r = 0; r < tl.numRounds; ++r) {
      const uint32_t b = roundOfs[tl.roundBegin + r], e = roundOfs[tl.roundBegin + r + 1u];
      for (uint32_t k = b + threadIdx.x; k < e; k += 256u) {
        const int4 it = items[tl.itemBegin + k];
        float c[2][3];
        const int32_t ref[2] = {it.y, it.z};
        for (int s = 0; s < 2; ++s) {
          if (ref[s] >= 0) { for (int q = 0; q < 3; ++q) c[s][q] = sF[ref[s]][q]; }
          else if ((uint32_t)ref[s] & 0x40000000u) { c[s][0] = c[s][1] = c[s][2] = 0.0f; }      // a leaf
          else { const float4 f = cost4[(uint32_t)ref[s] & 0x3FFFFFFFu]; c[s][0] = f.x; c[s][1] = f.y; c[s][2] = f.z; }      // a node of a lower treelet level
        }
        const float A = halfAreaOf(nodeBox + 6 * (size_t)it.x) * ka + (float)(cnt0[it.x] + 1u) * kt;
        float best = c[0][0] + c[1][2]; uint32_t k1 = 1u;
        { const float v = c[0][1] + c[1][1]; if (v < best) { best = v; k1 = 2u; } }
        { const float v = c[0][2] + c[1][0]; if (v < best) { best = v; k1 = 3u; } }
        const float F1 = A + best;
        float F2 = F1; uint32_t d2 = 0u;
        { const float v = c[0][0] + c[1][0]; if (v < F2) { F2 = v; d2 = 1u; } }
        float F3 = F1; uint32_t d3 = 0u;
        { const float v = c[0][0] + c[1][1]; if (v < F3) { F3 = v; d3 = 1u; } }
        { const float v = c[0][1] + c[1][0]; if (v < F3) { F3 = v; d3 = 2u; } }
        sF[k][0] = F1; sF[k][1] = F2; sF[k][2] = F3;
        cost4[it.x] = make_float4(F1, F2, F3, __uint_as_float(k1 | (d2 << 2) | (d3 << 4)));
      }
      __syncthreads();
    }
  }
}
__global__ void entries4Kernel(int numNodes, const int32_t* __restrict__ left, const int32_t* __restrict__ right, const float4* __restrict__ cost4, int4* __restrict__ ent4) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= numNodes) return;
  int32_t e[4] = {RT_BVH4_EMPTY, RT_BVH4_EMPTY, RT_BVH4_EMPTY, RT_BVH4_EMPTY};
  int cnt = 0;
  // (subtree, entries it may become) pairs still to settle, left to right; a pair with one entry, or a leaf, is an entry
  int32_t stN[4]; int stI[4]; int sp = 0;
  { const uint32_t k1 = __float_as_uint(cost4[i].w) & 3u; stN[0] = right[i]; stI[0] = 4 - (int)k1; stN[1] = left[i]; stI[1] = (int)k1; sp = 2; }
  while (sp > 0) {
    const int32_t n = stN[--sp]; const int allowed = stI[sp];
    if (n < 0 || allowed == 1) { e[cnt++] = n; continue; }
    const uint32_t w = __float_as_uint(cost4[n].w);
    const int k = allowed == 2 ? (int)((w >> 2) & 1u) : (int)((w >> 4) & 3u);      // 0: n stays one entry; else n is opened k : allowed - k
    if (k == 0) { e[cnt++] = n; continue; }
    stN[sp] = right[n]; stI[sp] = allowed - k; ++sp;
    stN[sp] = left[n]; stI[sp] = k; ++sp;
  }
  ent4[i] = make_int4(e[0], e[1], e[2], e[3]);
}
#define RT_MAX_TREE_DEPTH 128      // ancestors of a node roots4Kernel can hold (PLOC trees of 100 000 triangles: ~25; deeper: BuildResult::error bit 2)
RT_DEV int entryCount(const int4 e) { return 2 + (e.z != RT_BVH4_EMPTY ? 1 : 0) + (e.w != RT_BVH4_EMPTY ? 1 : 0); }
__global__ void __launch_bounds__(256) roots4Kernel(int numNodes, const int32_t* __restrict__ nodeParent, const int4* __restrict__ ent4, uint32_t* __restrict__ lvl4, BuildResult* res) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t myStack = 0, isRoot = 0, level = 0;
  if (i < numNodes) {
    int32_t path[RT_MAX_TREE_DEPTH];      // path[0] = i ... path[d - 1] = the root
    int d = 0;
    for (int32_t p = i; p >= 0; p = nodeParent[p]) { if (d < RT_MAX_TREE_DEPTH) path[d] = p; ++d; }
    if (d > RT_MAX_TREE_DEPTH) atomicOr(&res->error, 4u);
    else {
      uint32_t bound = 0;
      int j = d - 1;
      for (;;) {
        if (j == 0) { isRoot = 1u; break; }
        const int4 e = ent4[path[j]];
        int nj = -1;      // the next 4-wide node on the way down: the ancestor of i (or i) among this node's entries, at most three levels below
        for (int s = 1; s <= 3 && j - s >= 0; ++s) { const int32_t x = path[j - s]; if (x == e.x || x == e.y || x == e.z || x == e.w) { nj = j - s; break; } }
        if (nj < 0) break;      // i was opened: it lies inside this 4-wide node
        bound += (uint32_t)entryCount(e) - 1u; ++level; j = nj;
      }
      if (isRoot) myStack = bound + (uint32_t)entryCount(ent4[i]) - 1u;
    }
    lvl4[i] = isRoot ? level : 0xFFFFFFFFu;
  }
  // one atomic per workgroup and result (thousands of atomics on one word would take longer than the kernel)
  __shared__ uint32_t sStack, sCount, sLevel;
  if (threadIdx.x == 0) { sStack = 0u; sCount = 0u; sLevel = 0u; }
  __syncthreads();
  const unsigned long long rootMask = __ballot(isRoot != 0u);
  for (int o = 32; o > 0; o >>= 1) { myStack = max(myStack, (uint32_t)__shfl_down((int)myStack, o)); level = max(level, (uint32_t)__shfl_down((int)level, o)); }
  if ((threadIdx.x & 63) == 0) { atomicMax(&sStack, myStack); atomicMax(&sLevel, level); atomicAdd(&sCount, (uint32_t)__popcll(rootMask)); }
  __syncthreads();
  if (threadIdx.x == 0 && sCount) { atomicMax(&res->stack4, sStack); atomicMax(&res->depth4, sLevel + 1u); atomicAdd(&res->nodes4, sCount); }
}
// The 4-wide node of binary node i, if it is one: its entries' boxes from the binary tree's (triBox / nodeBox of the latest refit).
__global__ void emitNodes4(int n, const uint32_t* __restrict__ order, const float* __restrict__ triBox, const float* __restrict__ nodeBox,
                           const int4* __restrict__ ent4, const uint32_t* __restrict__ lvl4, Bvh4Node* __restrict__ nodes4) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1) return;
  const uint32_t level = lvl4[i];
  if (level == 0xFFFFFFFFu) return;
  const int4 e4 = ent4[i];
  const int32_t refs[4] = {e4.x, e4.y, e4.z, e4.w};
  Bvh4Node nd;
  for (int k = 0; k < 4; ++k) {
    if (refs[k] != RT_BVH4_EMPTY) {
      const float* b = refs[k] < 0 ? &triBox[6 * (size_t)order[~refs[k]]] : &nodeBox[6 * (size_t)refs[k]];
      nd.minx[k] = b[0]; nd.miny[k] = b[1]; nd.minz[k] = b[2]; nd.maxx[k] = b[3]; nd.maxy[k] = b[4]; nd.maxz[k] = b[5];
    } else {
      nd.minx[k] = nd.miny[k] = nd.minz[k] = __builtin_inff(); nd.maxx[k] = nd.maxy[k] = nd.maxz[k] = -__builtin_inff();
    }
    nd.ref[k] = refs[k];
    nd.pad[k] = 0;
  }
  nd.pad[0] = (int32_t)level;      // level in the 4-wide tree (read by -DRT_TRACE_STATS builds and tools/probes only)
  nodes4[i] = nd;
}
// Leaf slots in depth-first order of the finished tree.  PLOC starts from the Morton order, and the clusters it merges are neighbours
// within its search radius, not adjacent ones: a subtree's triangles are scattered over up to ~32 slots.  The rank of a leaf in a
// depth-first walk is the number of triangles to its left: on the way up to the root, the left sibling's subtree wherever the path
// arrives from the right (cnt0: internal nodes below a node = its triangles - 1).  leafPermuteKernel then moves every per-slot array
// to the new slots and rewrites the leaf references of the nodes.
__global__ void leafRankKernel(int n, const int32_t* __restrict__ left, const int32_t* __restrict__ right, const int32_t* __restrict__ nodeParent,
                               const int32_t* __restrict__ leafParent, const uint32_t* __restrict__ cnt0, uint32_t* __restrict__ rank) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  uint32_t r = 0;
  int32_t ref = ~s;
  for (int32_t cur = leafParent[s]; cur >= 0; cur = nodeParent[cur]) {
    const int32_t l = left[cur];
    if (right[cur] == ref) r += l < 0 ? 1u : cnt0[l] + 1u;
    ref = cur;
  }
  rank[s] = r;
}
__global__ void leafPermuteKernel(int n, const uint32_t* __restrict__ rank, const uint32_t* __restrict__ orderIn, const int32_t* __restrict__ leafParentIn,
                                  uint32_t* __restrict__ orderOut, int32_t* __restrict__ leafParentOut, int32_t* __restrict__ left, int32_t* __restrict__ right) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const uint32_t r = rank[i]; orderOut[r] = orderIn[i]; leafParentOut[r] = leafParentIn[i]; }
  if (i < n - 1) {
    const int32_t l = left[i], rr = right[i];
    if (l < 0) left[i] = ~(int32_t)rank[~l];
    if (rr < 0) right[i] = ~(int32_t)rank[~rr];
  }
}
__global__ void emitTris(int n, const uint32_t* __restrict__ order, const float* __restrict__ verts, const uint32_t* __restrict__ idx, BvhTri* __restrict__ tris) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const uint32_t prim = order[s];
  BvhTri t;
  for (int k = 0; k < 3; ++k) {
    t.v0[k] = verts[6 * (size_t)idx[3 * (size_t)prim] + k];
    t.v1[k] = verts[6 * (size_t)idx[3 * (size_t)prim + 1] + k];
    t.v2[k] = verts[6 * (size_t)idx[3 * (size_t)prim + 2] + k];
  }
  t.prim = prim;
  for (int k = 0; k < 3; ++k) { t.pad0[k] = 0; t.pad1[k] = 0; }
  t.pad0[0] = prim;      // once more in the record's third 16-byte word: the trace kernel fetches three words per triangle, not four
  tris[s] = t;
}
// The table of the tree's top for the trace kernel's LDS (rtggx_device.h RT_TOP_*): entry k = the 4-wide node topList[k], its
// references to nodes that are in the table themselves replaced by RT_TOP_FLAG | rank.
__global__ void emitTop(int count, const int32_t* __restrict__ topList, const int32_t* __restrict__ topRank, const Bvh4Node* __restrict__ nodes4, Bvh4Node* __restrict__ top) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= count) return;
  Bvh4Node nd = nodes4[topList[k]];
  for (int e = 0; e < 4; ++e) {
    const int32_t r = nd.ref[e];
    if (r >= 0 && r != RT_BVH4_EMPTY && topRank[r] >= 0) nd.ref[e] = RT_TOP_FLAG | topRank[r];
  }
  top[k] = nd;
}

// ---- the list of nodes at the tree's top (for the trace kernel's LDS table), on the device -----------------------------------------
// The first `capacity` 4-wide nodes in breadth-first order: their list and every node's rank in it (topRank: -1 elsewhere, cleared by
// the caller).  One workgroup walks the tree level by level through the 4-wide entries (ent4); the order is the one a queue would give:
// a node's internal entries in entry order, nodes in list order.
__global__ void __launch_bounds__(128) planTopKernel(uint32_t capacity, int32_t root, uint32_t numNodes, const int4* __restrict__ ent4,
                                                     int32_t* __restrict__ topList, int32_t* __restrict__ topRank, BuildResult* res) {
  __shared__ int32_t list[128]; __shared__ uint32_t offs[128]; __shared__ uint32_t sCount, sHead;
  if (capacity > 128u) capacity = 128u;
  if (threadIdx.x == 0) { sCount = 0u; sHead = 0u; if (root >= 0 && numNodes > 0u && capacity > 0u) { list[0] = root; sCount = 1u; } }
  __syncthreads();
  for (;;) {
    const uint32_t head = sHead, count = sCount;
    if (head >= count || count >= capacity) break;
    // the (at most four) entries of my node that are 4-wide nodes themselves, in entry order; -1: none
    int32_t g[4] = {-1, -1, -1, -1};
    if (head + threadIdx.x < count) {
      const int4 e = ent4[list[head + threadIdx.x]];
      const int32_t r[4] = {e.x, e.y, e.z, e.w};
      for (int k = 0; k < 4; ++k) if (r[k] >= 0 && r[k] != RT_BVH4_EMPTY) g[k] = r[k];
    }
    const uint32_t k = (g[0] >= 0 ? 1u : 0u) + (g[1] >= 0 ? 1u : 0u) + (g[2] >= 0 ? 1u : 0u) + (g[3] >= 0 ? 1u : 0u);
    offs[threadIdx.x] = k;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t run = 0; for (uint32_t t = 0; t < count - head; ++t) { const uint32_t v = offs[t]; offs[t] = run; run += v; } sHead = count; sCount = min(capacity, count + run); }
    __syncthreads();
    {
      uint32_t pos = count + offs[threadIdx.x];
      for (int q = 0; q < 4; ++q) if (g[q] >= 0) { if (pos < capacity) list[pos] = g[q]; ++pos; }
    }
    __syncthreads();
  }
  const uint32_t count = sCount;
  if (threadIdx.x < count) { topList[threadIdx.x] = list[threadIdx.x]; topRank[list[threadIdx.x]] = (int32_t)threadIdx.x; }
  if (threadIdx.x == 0) res->topCount = count;
}

// ---- the refit schedule, on the device ("treelets") -------------------------------------------------------------------------------
// A bottom-up box refit needs a node's children before the node.  The tree is cut into TREELETS of at most RT_TREELET_NODES internal
// nodes; one workgroup refits one treelet round by round of the PLOC build (a node's children belong to earlier rounds) with the
// treelet's boxes in LDS and a workgroup barrier between rounds; what is left above the treelet roots is cut the same way again
// (level 1, level 2): two launches for the bunny and the dragon, no global synchronisation (refitTreelets below).
// The cut needs, per node and level, how many nodes of its subtree that level still has to place: cnt[l] of plocEmit.  A node is
// PENDING at level l if level l - 1 could not place it (cnt[l-1] > RT_TREELET_NODES; at level 0 every node is pending); it ROOTS a
// treelet of level l if it is pending, its count fits, and its parent's does not.  The treelet is the pending part of its subtree.
// An item of a treelet: (node, left ref, right ref); a ref is >= 0: position of another item of the same treelet (its box is in LDS);
// 0xC0000000 | s: leaf slot s (the primitive's box); 0x80000000 | n: node n placed by a lower level (its box is final in nodeBox).
__global__ void treeletRootsKernel(int numNodes, int level, const uint32_t* __restrict__ cntPrev, const uint32_t* __restrict__ cntCur, const int32_t* __restrict__ nodeParent,
                                   int32_t* __restrict__ roots, BuildResult* res) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= numNodes) return;
  if (level > 0 && cntPrev[i] <= RT_TREELET_NODES) return;      // placed by a lower level
  const int32_t p = nodeParent[i];
  if (cntCur[i] > RT_TREELET_NODES) { if (p < 0 && level == RT_TREELET_LEVELS - 1) atomicOr(&res->error, 2u); return; }
  if (p >= 0 && cntCur[p] <= RT_TREELET_NODES) return;           // an inner node of somebody else's treelet
  roots[treeletBase(res, level) + atomicAdd(&res->treelets[level], 1u)] = i;
}
// The treelet below each root of `level`: its pending nodes breadth first, stored in REVERSE -- children before parents --, one round
// per level of the walk (deepest first).  (The first version sorted the nodes by index = by PLOC round: a bitonic sort and two binary
// searches per item; the levels of the walk order the nodes just as well.)
__global__ void __launch_bounds__(256) buildTreeletsKernel(int level, const uint32_t* __restrict__ cntPrev, const int32_t* __restrict__ left, const int32_t* __restrict__ right,
                                                           const int32_t* __restrict__ roots,
                                                           RefitTreelet* __restrict__ treelets, int4* __restrict__ items, uint32_t* __restrict__ rounds, BuildResult* res) {
  __shared__ int32_t list[RT_TREELET_NODES], refL[RT_TREELET_NODES], refR[RT_TREELET_NODES];
  __shared__ uint32_t levelStart[RT_TREELET_NODES + 1];
  __shared__ uint32_t sCount, sHead, sLevels, sItemBegin, sRoundBegin;
  const uint32_t base = treeletBase(res, level), K = res->treelets[level];
  const auto pending = [&](int32_t x) { return level == 0 || cntPrev[x] > RT_TREELET_NODES; };
  for (uint32_t k = blockIdx.x; k < K; k += gridDim.x) {
    if (threadIdx.x == 0) { list[0] = roots[base + k]; sCount = 1u; sHead = 0u; sLevels = 0u; }
    __syncthreads();
    for (;;) {
      const uint32_t head = sHead, count = min(sCount, (uint32_t)RT_TREELET_NODES);
      if (head >= count) break;
      __syncthreads();      // everybody has read head / count
      if (threadIdx.x == 0) levelStart[sLevels++] = head;
      for (uint32_t idx = head + threadIdx.x; idx < count; idx += 256u) {
        const int32_t v = list[idx];
        const int32_t lc = left[v], rc = right[v];
        // a child reference: < 0 as it will be stored (leaf / node of a lower level); >= 0 the child's position in the walk (turned round below)
        int32_t a = lc < 0 ? (int32_t)(0xC0000000u | (uint32_t)~lc) : (int32_t)(0x80000000u | (uint32_t)lc);
        int32_t b = rc < 0 ? (int32_t)(0xC0000000u | (uint32_t)~rc) : (int32_t)(0x80000000u | (uint32_t)rc);
        if (lc >= 0 && pending(lc)) { const uint32_t q = atomicAdd(&sCount, 1u); if (q < RT_TREELET_NODES) { list[q] = lc; a = (int32_t)q; } }
        if (rc >= 0 && pending(rc)) { const uint32_t q = atomicAdd(&sCount, 1u); if (q < RT_TREELET_NODES) { list[q] = rc; b = (int32_t)q; } }
        refL[idx] = a; refR[idx] = b;
      }
      __syncthreads();
      if (threadIdx.x == 0) sHead = count;
      __syncthreads();
    }
    const uint32_t count = min(sCount, (uint32_t)RT_TREELET_NODES), levels = sLevels;
    if (threadIdx.x == 0) { levelStart[levels] = count; sItemBegin = atomicAdd(&res->itemCursor, count); sRoundBegin = atomicAdd(&res->roundCursor, levels + 1u); }
    __syncthreads();
    for (uint32_t idx = threadIdx.x; idx < count; idx += 256u) {
      const int32_t a = refL[idx], b = refR[idx];
      items[sItemBegin + (count - 1u - idx)] = make_int4(list[idx], a >= 0 ? (int32_t)(count - 1u) - a : a, b >= 0 ? (int32_t)(count - 1u) - b : b, 0);
    }
    for (uint32_t rr = threadIdx.x; rr <= levels; rr += 256u) rounds[sRoundBegin + rr] = count - levelStart[levels - rr];      // round rr = level (levels - 1 - rr)
    if (threadIdx.x == 0) { RefitTreelet tl; tl.itemBegin = sItemBegin; tl.roundBegin = sRoundBegin; tl.numRounds = levels; tl.pad = 0u; treelets[base + k] = tl; }
    __syncthreads();
  }
}

// ---- refit (rtggx_refit_as): same topology, new vertex positions ---------------------------------------------------------
// Leaf slot s holds primitive order[s] (the topology's; the record may still hold another topology's): its three vertices and the primitive's box.
__global__ void refitTris(int n, const uint32_t* __restrict__ order, const float* __restrict__ verts, const uint32_t* __restrict__ idx, BvhTri* __restrict__ tris, float* __restrict__ triBox) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const uint32_t prim = order[s];
  float mn[3], mx[3];
  BvhTri t;
  t.prim = prim;
  for (int k = 0; k < 3; ++k) { t.pad0[k] = 0; t.pad1[k] = 0; }
  t.pad0[0] = prim;
  for (int k = 0; k < 3; ++k) {
    const float a = verts[6 * (size_t)idx[3 * (size_t)prim] + k], b = verts[6 * (size_t)idx[3 * (size_t)prim + 1] + k], c = verts[6 * (size_t)idx[3 * (size_t)prim + 2] + k];
    t.v0[k] = a; t.v1[k] = b; t.v2[k] = c;
    mn[k] = fminf(a, fminf(b, c)); mx[k] = fmaxf(a, fmaxf(b, c));
  }
  tris[s] = t;
  for (int k = 0; k < 3; ++k) { triBox[6 * (size_t)prim + k] = mn[k]; triBox[6 * (size_t)prim + 3 + k] = mx[k]; }
}
// Sum of the half-areas of all node boxes: the tree's SAH cost up to constants.  A refit keeps the topology the build chose for
// the OLD shape; when this sum has grown by RT_REFIT_REBUILD_RATIO the host rebuilds (capi.hip).
__global__ void __launch_bounds__(256) treeCostKernel(int numNodes, const float* __restrict__ nodeBox, float* __restrict__ cost) {
  __shared__ float red[4];
  float a = 0.0f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < numNodes; i += gridDim.x * 256) {      // (few workgroups: the atomics on one word are what this kernel costs)
    const float* b = nodeBox + 6 * (size_t)i;
    const float ex = b[3] - b[0], ey = b[4] - b[1], ez = b[5] - b[2];
    a += (ex * ey + ey * ez) + ez * ex;
  }
  for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) { const float t = (red[0] + red[1]) + (red[2] + red[3]); if (t != 0.0f) atomicAdd(cost, t); }
}

// One workgroup refits one treelet (see "the refit schedule" above): round by round, the treelet's boxes in LDS.
__global__ void __launch_bounds__(256) refitTreelets(const RefitTreelet* __restrict__ treelets, const int4* __restrict__ items, const uint32_t* __restrict__ roundOfs,
                                                     const uint32_t* __restrict__ order, const float* __restrict__ triBox, float* __restrict__ nodeBox) {
  __shared__ float box[RT_TREELET_NODES][6];
  const RefitTreelet tl = treelets[blockIdx.x];
  for (uint32_t r = 0; r < tl.numRounds; ++r) {
    const uint32_t b = roundOfs[tl.roundBegin + r], e = roundOfs[tl.roundBegin + r + 1u];
    for (uint32_t k = b + threadIdx.x; k < e; k += 256u) {
      const int4 it = items[tl.itemBegin + k];
      float c[2][6];
      const int32_t ref[2] = {it.y, it.z};
      for (int s = 0; s < 2; ++s) {
        if (ref[s] >= 0) { for (int q = 0; q < 6; ++q) c[s][q] = box[ref[s]][q]; }
        else {
          const uint32_t x = (uint32_t)ref[s] & 0x3FFFFFFFu;
          const float* p = ((uint32_t)ref[s] & 0x40000000u) ? &triBox[6 * (size_t)order[x]] : &nodeBox[6 * (size_t)x];
          for (int q = 0; q < 6; ++q) c[s][q] = p[q];
        }
      }
      for (int q = 0; q < 3; ++q) {
        const float mn = fminf(c[0][q], c[1][q]), mx = fmaxf(c[0][3 + q], c[1][3 + q]);
        box[k][q] = mn; box[k][3 + q] = mx;
        nodeBox[6 * (size_t)it.x + q] = mn; nodeBox[6 * (size_t)it.x + 3 + q] = mx;
      }
    }
    __syncthreads();
  }
}

// Fat triangles (rtggx_context.h): primitive p's three vertices, 6 floats each, at fat[5 p .. 5 p + 4].
__global__ void emitFatTris(uint32_t n, const float* __restrict__ verts, const uint32_t* __restrict__ idx, float4* __restrict__ fat) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  float v[20];
  for (int k = 0; k < 3; ++k) { const float* src = verts + 6 * (size_t)idx[3 * (size_t)p + k]; for (int q = 0; q < 6; ++q) v[6 * k + q] = src[q]; }
  v[18] = 0.0f; v[19] = 0.0f;
  for (int q = 0; q < 5; ++q) fat[5 * (size_t)p + q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}
int buildFatTris(rtggx_context* c, uint32_t slot, uint32_t set, hipStream_t s) {
  MeshDev& m = c->mesh[slot];
  if (m.numTris == 0 || !m.fatBuf[set]) return 0;
  hipLaunchKernelGGL(emitFatTris, dim3((m.numTris + 255) / 256), dim3(256), 0, s, m.numTris, (const float*)m.vertsBuf[set], (const uint32_t*)m.indices, m.fatBuf[set]);
  RT_HIP(hipGetLastError());
  return 0;
}

// Buffers that exist once per input set when the mesh deforms and alias one allocation while it is static.
template <typename T> static void freeAliased(T* (&buf)[RT_SETS]) {
  for (int i = 0; i < RT_SETS; ++i) { bool dup = false; for (int j = 0; j < i; ++j) dup = dup || buf[j] == buf[i]; if (!dup && buf[i]) hipFree(buf[i]); }
  for (auto& b : buf) b = nullptr;
}
static void freeTopo(BvhTopo& t) {
  hipFree(t.order); hipFree(t.left); hipFree(t.right); hipFree(t.nodeParent); hipFree(t.leafParent); hipFree(t.nodeBox); hipFree(t.triBox);
  for (auto& c : t.cnt) { hipFree(c); c = nullptr; }
  hipFree(t.roundBase); hipFree(t.dTreelets); hipFree(t.dRefitItems); hipFree(t.dRefitRounds); hipFree(t.treeletRoots); hipFree(t.topList); hipFree(t.topRank); hipFree(t.ent4); hipFree(t.lvl4); hipFree(t.cost4); hipFree(t.dResult);
  if (t.hResult) hipHostFree(t.hResult);
  t = BvhTopo{};
}

// ---- a build as a list of launches ------------------------------------------------------------------------------------------------
struct BuildScratch {
  uint32_t *codes[2] = {}, *order2 = nullptr, *hist = nullptr, *chunkSums = nullptr, *bounds = nullptr;
  int32_t *clRef[2] = {}, *nn = nullptr; float* clBox[2] = {}; uint2* blockCounts = nullptr; PlocState* state = nullptr;
  float* vertsSnapshot = nullptr;      // a rebuild beside the frames works on a copy of the vertices it started from
};
struct BuildJob {
  BvhTopo topo;                        // what the build produces (swapped with the mesh's when it has ended)
  BuildScratch s;
  std::vector<std::function<void(hipStream_t)>> steps;      // one kernel launch (or copy) each
  size_t next = 0;
  hipEvent_t done = nullptr;
  bool active = false, allIssued = false;
  uint32_t numTris = 0, numVerts = 0;
};
static void freeScratch(BuildScratch& s) {
  hipFree(s.codes[0]); hipFree(s.codes[1]); hipFree(s.order2); hipFree(s.hist); hipFree(s.chunkSums); hipFree(s.bounds);
  hipFree(s.clRef[0]); hipFree(s.clRef[1]); hipFree(s.nn); hipFree(s.clBox[0]); hipFree(s.clBox[1]); hipFree(s.blockCounts); hipFree(s.state); hipFree(s.vertsSnapshot);
  s = BuildScratch{};
}
void freeBuildProducts(MeshDev& m) {
  if (m.job) { if (m.job->done) hipEventDestroy(m.job->done); freeTopo(m.job->topo); freeScratch(m.job->s); delete m.job; m.job = nullptr; }
  freeTopo(m.topo);
  freeAliased(m.nodesBuf); freeAliased(m.nodes4Buf); freeAliased(m.trisBuf); freeAliased(m.topBuf);
  m.nodes = nullptr; m.nodes4 = nullptr; m.tris = nullptr; m.top = nullptr; m.topCount = 0;
  for (auto& t : m.topCountBuf) t = 0;
}
static int allocTopo(BvhTopo& t, uint32_t n) {
  const size_t nn = n > 1 ? n : 2;
  t.numTris = n;
  RT_HIP(hipMalloc(&t.order, 4 * nn)); RT_HIP(hipMalloc(&t.left, 4 * nn)); RT_HIP(hipMalloc(&t.right, 4 * nn));
  RT_HIP(hipMalloc(&t.nodeParent, 4 * nn)); RT_HIP(hipMalloc(&t.leafParent, 4 * nn));
  RT_HIP(hipMalloc(&t.nodeBox, 24 * nn)); RT_HIP(hipMalloc(&t.triBox, 24 * nn));
  for (auto& c : t.cnt) RT_HIP(hipMalloc(&c, 4 * nn));
  RT_HIP(hipMalloc(&t.roundBase, 4 * (RT_MAX_ROUNDS + 1)));
  RT_HIP(hipMalloc(&t.dTreelets, sizeof(RefitTreelet) * nn)); RT_HIP(hipMalloc(&t.dRefitItems, sizeof(int4) * nn)); RT_HIP(hipMalloc(&t.dRefitRounds, 4 * (2 * nn + 64)));
  RT_HIP(hipMalloc(&t.treeletRoots, 4 * nn));
  RT_HIP(hipMalloc(&t.topList, 4 * 128)); RT_HIP(hipMalloc(&t.topRank, 4 * nn));
  RT_HIP(hipMalloc(&t.ent4, sizeof(int4) * nn)); RT_HIP(hipMalloc(&t.lvl4, 4 * nn)); RT_HIP(hipMalloc(&t.cost4, sizeof(float4) * nn));
  RT_HIP(hipMalloc(&t.dResult, sizeof(BuildResult))); RT_HIP(hipHostMalloc(&t.hResult, sizeof(BuildResult)));
  memset(t.hResult, 0, sizeof(BuildResult));
  return 0;
}
static int allocScratch(BuildScratch& s, uint32_t n, uint32_t nv, bool snapshot) {
  const size_t nn = n > 1 ? n : 2, nb = (n + 255) / 256;
  RT_HIP(hipMalloc(&s.codes[0], 4 * nn)); RT_HIP(hipMalloc(&s.codes[1], 4 * nn)); RT_HIP(hipMalloc(&s.order2, 4 * nn));
  RT_HIP(hipMalloc(&s.hist, 4 * 256 * nb)); RT_HIP(hipMalloc(&s.chunkSums, 4 * ((256 * nb + 1023) / 1024 + 1))); RT_HIP(hipMalloc(&s.bounds, 4 * 8));
  RT_HIP(hipMalloc(&s.clRef[0], 4 * nn)); RT_HIP(hipMalloc(&s.clRef[1], 4 * nn)); RT_HIP(hipMalloc(&s.nn, 4 * nn));
  RT_HIP(hipMalloc(&s.clBox[0], 24 * nn)); RT_HIP(hipMalloc(&s.clBox[1], 24 * nn));
  RT_HIP(hipMalloc(&s.blockCounts, sizeof(uint2) * nb)); RT_HIP(hipMalloc(&s.state, 2 * sizeof(PlocState)));
  if (snapshot) RT_HIP(hipMalloc(&s.vertsSnapshot, sizeof(float) * 6 * (size_t)nv));
  return 0;
}
// How many multi-workgroup rounds the host issues: until RT_PLOC_STOP clusters are expected to be left if every round kept RT_PLOC_KEEP of
// its clusters (measured, RTGGX_BUILD_LOG: the bunny keeps ~0.77 per round, the dragon ~0.835).  A round that is not needed returns at
// once (plocNearest); a mesh that keeps more reaches plocFinal with more clusters -- up to RT_PLOC_LDS of them go straight into LDS,
// beyond that it starts with rounds on the lists in global memory.  (Round 3's first version budgeted for 0.77 + one round and 2048
// clusters in LDS: the dragon's plocFinal then started with 5523 clusters and took 481 us of the build's 881.)
static uint32_t plocRoundsFor(uint32_t n) {
  uint32_t rounds = 0; double m = (double)n;
  while (m > (double)RT_PLOC_STOP) { m *= RT_PLOC_KEEP; ++rounds; }
  return rounds;
}

// The launches of one build of mesh `slot` from `verts` (device; the job's snapshot when `snapshotFrom` is given), into job.topo.
static void planBuildSteps(rtggx_context* c, uint32_t slot, BuildJob& job, const float* verts, const float* snapshotFrom) {
  MeshDev& m = c->mesh[slot];
  const uint32_t n = job.numTris, nv = job.numVerts, nb = (n + 255) / 256;
  BvhTopo& t = job.topo; BuildScratch& s = job.s;
  auto& steps = job.steps;
  steps.clear(); job.next = 0; job.allIssued = false;
  const uint32_t* indices = m.indices;
  if (snapshotFrom) { steps.push_back([=](hipStream_t st) { hipMemcpyAsync(s.vertsSnapshot, snapshotFrom, sizeof(float) * 6 * (size_t)nv, hipMemcpyDeviceToDevice, st); }); verts = s.vertsSnapshot; }
  steps.push_back([=](hipStream_t st) { hipLaunchKernelGGL(buildBegin, dim3(1), dim3(64), 0, st, n, s.state, t.dResult, s.bounds); });
  steps.push_back([=](hipStream_t st) { hipLaunchKernelGGL(boundsKernel, dim3(std::min<uint32_t>((nv + 255) / 256, 64u)), dim3(256), 0, st, verts, nv, s.bounds); });
  steps.push_back([=](hipStream_t st) { hipLaunchKernelGGL(mortonKernel, dim3(nb), dim3(256), 0, st, verts, indices, n, (const uint32_t*)s.bounds, s.codes[0], t.order, t.triBox); });
  // radix sort: (codes[0], t.order) <-> (codes[1], s.order2); four passes end where they began
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = pass * 8, cur = pass & 1;
    uint32_t *kin = s.codes[cur], *kout = s.codes[cur ^ 1], *vin = cur ? s.order2 : t.order, *vout = cur ? t.order : s.order2;
    steps.push_back([=](hipStream_t st) { hipLaunchKernelGGL(radixHist, dim3(nb), dim3(256), 0, st, (const uint32_t*)kin, n, shift, s.hist, nb); });
    const uint32_t chunks = (256u * nb + 1023u) / 1024u;
    steps.push_back([=](hipStream_t st) { hipLaunchKernelGGL(scanChunks, dim3(chunks), dim3(1024), 0, st, s.hist, 256u * nb, s.chunkSums); });
    steps.push_back([=](hipStream_t st) { hipLaunchKernelGGL(scanTotals, dim3(1), dim3(1024), 0, st, s.chunkSums, chunks); });
    steps.push_back([=](hipStream_t st) { hipLaunchKernelGGL(radixScatter, dim3(nb), dim3(256), 0, st, (const uint32_t*)kin, (const uint32_t*)vin, n, shift, (const uint32_t*)s.hist, (const uint32_t*)s.chunkSums, nb, kout, vout); });
  }
  t.root = n == 1 ? ~0 : -1; t.refittable = false;
  if (n > 1) {
    const int radius = RT_PLOC_RADIUS;
    {
      t.root = (int32_t)n - 2; t.refittable = true;      // the last node created
      PlocArrays A;
      A.clRef[0] = s.clRef[0]; A.clRef[1] = s.clRef[1]; A.clBox[0] = s.clBox[0]; A.clBox[1] = s.clBox[1]; A.nn = s.nn;
      A.left = t.left; A.right = t.right; A.nodeParent = t.nodeParent; A.leafParent = t.leafParent; A.nodeBox = t.nodeBox;
      for (int l = 0; l < RT_TREELET_LEVELS; ++l) A.cnt[l] = t.cnt[l];
      A.roundBase = t.roundBase; A.res = t.dResult;
      steps.push_back([=](hipStream_t st) { hipLaunchKernelGGL(plocInit, dim3(nb), dim3(256), 0, st, (int)n, (const uint32_t*)t.order, (const float*)t.triBox, s.clRef[0], s.clBox[0]); });
      const uint32_t rounds = plocRoundsFor(n);
      for (uint32_t r = 0; r < rounds; ++r) {
        steps.push_back([=](hipStream_t st) { hipLaunchKernelGGL(plocNearest, dim3(nb), dim3(256 * RT_PLOC_LANES), 0, st, (const PlocState*)s.state, r, radius, A, s.blockCounts); });
        steps.push_back([=](hipStream_t st) { hipLaunchKernelGGL(plocScatter, dim3(nb), dim3(256), 0, st, s.state, r, (const uint2*)s.blockCounts, A); });
      }
      steps.push_back([=](hipStream_t st) { hipLaunchKernelGGL(plocFinal, dim3(1), dim3(1024), 0, st, (const PlocState*)s.state, rounds, radius, n, A); });
      // leaf slots in depth-first order (a subtree's triangles are consecutive); scratch: codes[1] = ranks, order2 / clRef[0] = the moved arrays
      steps.push_back([=](hipStream_t st) { hipLaunchKernelGGL(leafRankKernel, dim3(nb), dim3(256), 0, st, (int)n, (const int32_t*)t.left, (const int32_t*)t.right, (const int32_t*)t.nodeParent, (const int32_t*)t.leafParent,
                                                               (const uint32_t*)t.cnt[0], s.codes[1]); });
      steps.push_back([=](hipStream_t st) { hipLaunchKernelGGL(leafPermuteKernel, dim3(nb), dim3(256), 0, st, (int)n, (const uint32_t*)s.codes[1], (const uint32_t*)t.order, (const int32_t*)t.leafParent,
                                                               s.order2, s.clRef[0], t.left, t.right); });
      steps.push_back([=](hipStream_t st) { hipMemcpyAsync(t.order, s.order2, 4 * (size_t)n, hipMemcpyDeviceToDevice, st); });
      steps.push_back([=](hipStream_t st) { hipMemcpyAsync(t.leafParent, s.clRef[0], 4 * (size_t)n, hipMemcpyDeviceToDevice, st); });
      // the refit schedule
      const uint32_t treeletGrid = std::min<uint32_t>(std::max<uint32_t>((n + RT_TREELET_NODES / 4 - 1) / (RT_TREELET_NODES / 4), 1u), 2048u);
      for (int l = 0; l < RT_TREELET_LEVELS; ++l) {
        const uint32_t* prev = l ? t.cnt[l - 1] : nullptr; const uint32_t* cur = t.cnt[l];
        steps.push_back([=](hipStream_t st) { hipLaunchKernelGGL(treeletRootsKernel, dim3(nb), dim3(256), 0, st, (int)n - 1, l, prev, cur, (const int32_t*)t.nodeParent, t.treeletRoots, t.dResult); });
        steps.push_back([=](hipStream_t st) { hipLaunchKernelGGL(buildTreeletsKernel, dim3(l ? 64u : treeletGrid), dim3(256), 0, st, l, prev, (const int32_t*)t.left, (const int32_t*)t.right,
                                                                 (const int32_t*)t.treeletRoots, (RefitTreelet*)t.dTreelets, (int4*)t.dRefitItems, t.dRefitRounds, t.dResult); });
      }
    }
    const uint32_t topCap = slot == 0 ? RT_TOP_SLOT0 : RT_TOP_SLOT1;
    const int32_t root = t.root;
    steps.push_back([=](hipStream_t st) { hipMemsetAsync(t.topRank, 0xFF, 4 * (size_t)(n - 1), st); });
    // the 4-wide collapse (by surface area), then the table of its top
    const float wArea = c->collapseWeights[0], wTris = c->collapseWeights[1];
    for (int l = 0; l < RT_TREELET_LEVELS; ++l)
      steps.push_back([=](hipStream_t st) { hipLaunchKernelGGL(collapseCostTreelets, dim3(l ? 64u : std::min<uint32_t>(std::max<uint32_t>((n + RT_TREELET_NODES / 4 - 1) / (RT_TREELET_NODES / 4), 1u), 2048u)), dim3(256), 0, st, l,
                                                               (const RefitTreelet*)t.dTreelets, (const int4*)t.dRefitItems, (const uint32_t*)t.dRefitRounds, (const float*)t.nodeBox, (const uint32_t*)t.cnt[0], root, wArea, wTris, t.cost4, (const BuildResult*)t.dResult); });
    steps.push_back([=](hipStream_t st) { hipLaunchKernelGGL(entries4Kernel, dim3(nb), dim3(256), 0, st, (int)n - 1, (const int32_t*)t.left, (const int32_t*)t.right, (const float4*)t.cost4, t.ent4); });
    steps.push_back([=](hipStream_t st) { hipLaunchKernelGGL(roots4Kernel, dim3(nb), dim3(256), 0, st, (int)n - 1, (const int32_t*)t.nodeParent, (const int4*)t.ent4, t.lvl4, t.dResult); });
    steps.push_back([=](hipStream_t st) { hipLaunchKernelGGL(planTopKernel, dim3(1), dim3(128), 0, st, topCap, root, n - 1, (const int4*)t.ent4, t.topList, t.topRank, t.dResult); });
    steps.push_back([=](hipStream_t st) { hipLaunchKernelGGL(depthKernel, dim3(nb), dim3(256), 0, st, (int)n, (const int32_t*)t.nodeParent, (const int32_t*)t.leafParent, &t.dResult->depth); });
    steps.push_back([=](hipStream_t st) { hipLaunchKernelGGL(treeCostKernel, dim3(std::min<uint32_t>((n - 1 + 255) / 256, 48u)), dim3(256), 0, st, (int)n - 1, (const float*)t.nodeBox, &t.dResult->cost); });
  }
  steps.push_back([=](hipStream_t st) { hipMemcpyAsync(t.hResult, t.dResult, sizeof(BuildResult), hipMemcpyDeviceToHost, st); });
}

// The tree arrays of input set `set` from a topology and that set's vertices: leaf triangles, 64-byte binary nodes (for the oracle /
// tests), their 4-wide collapse (for the trace kernel), the table of the tree's top.  `full`: the arrays held another topology (or
// nothing) before: unused entries of the sparse 4-wide array are cleared.
static void emitTree(MeshDev& m, const BvhTopo& t, uint32_t set, bool full, hipStream_t s) {
  const uint32_t n = t.numTris, nb = (n + 255) / 256;
  if (n > 1) {
    if (full) hipMemsetAsync(m.nodes4Buf[set], 0, sizeof(Bvh4Node) * (size_t)(n - 1), s);
    hipLaunchKernelGGL(emitNodes, dim3(nb), dim3(256), 0, s, (int)n, (const uint32_t*)t.order, (const float*)t.triBox, (const int32_t*)t.left, (const int32_t*)t.right, (const float*)t.nodeBox, m.nodesBuf[set]);
    hipLaunchKernelGGL(emitNodes4, dim3(nb), dim3(256), 0, s, (int)n, (const uint32_t*)t.order, (const float*)t.triBox, (const float*)t.nodeBox, (const int4*)t.ent4, (const uint32_t*)t.lvl4, m.nodes4Buf[set]);
    if (t.result.topCount) hipLaunchKernelGGL(emitTop, dim3((t.result.topCount + 63) / 64), dim3(64), 0, s, (int)t.result.topCount, (const int32_t*)t.topList, (const int32_t*)t.topRank, (const Bvh4Node*)m.nodes4Buf[set], m.topBuf[set]);
  }
  m.topCountBuf[set] = n > 1 ? t.result.topCount : 0u;
}

static int harvest(MeshDev& m, BvhTopo& t, uint32_t slot) {
  t.result = *t.hResult;
  if (t.result.error & 1u) { setError("BVH build of mesh %u: more than %u clustering rounds", slot, (unsigned)RT_MAX_ROUNDS); return -3; }
  if (t.result.error & 4u) { setError("BVH build of mesh %u: the tree is deeper than %d levels", slot, RT_MAX_TREE_DEPTH); return -3; }
  if (t.result.error & 2u) { setError("BVH build of mesh %u: the refit schedule would need more than %d treelet levels", slot, RT_TREELET_LEVELS); return -3; }
  if (t.result.depth > m.depth) m.depth = t.result.depth;
  m.stack4 = t.result.stack4;
  static const bool log = getenv("RTGGX_BUILD_LOG") != nullptr;
  if (log) fprintf(stderr, "[rtggx] build of mesh %u: %u triangles, %u rounds (%u clusters handed to the last workgroup, %u at the start of its LDS rounds), treelets %u / %u / %u, table %u, depth %u, %u 4-wide nodes in %u levels (deepest traversal stack %u), cost %.1f\n",
                   slot, t.numTris, t.result.numRounds, t.result.finalEntry, t.result.finalLds, t.result.treelets[0], t.result.treelets[1], t.result.treelets[2], t.result.topCount, t.result.depth,
                   t.result.nodes4, t.result.depth4, t.result.stack4, t.result.cost);
  if (log && t.numTris > 1 && t.numTris <= 64) {      // small trees: the arrays themselves
    const uint32_t nn = t.numTris - 1;
    std::vector<int32_t> l(nn), r(nn), tl(128), tr(nn); std::vector<uint32_t> c0(nn);
    hipMemcpy(l.data(), t.left, 4 * nn, hipMemcpyDeviceToHost); hipMemcpy(r.data(), t.right, 4 * nn, hipMemcpyDeviceToHost); hipMemcpy(c0.data(), t.cnt[0], 4 * nn, hipMemcpyDeviceToHost);
    hipMemcpy(tl.data(), t.topList, 4 * 128, hipMemcpyDeviceToHost); hipMemcpy(tr.data(), t.topRank, 4 * nn, hipMemcpyDeviceToHost);
    for (uint32_t i = 0; i < nn; ++i) fprintf(stderr, "   node %u: left %d right %d cnt0 %u topRank %d\n", i, l[i], r[i], c0[i], tr[i]);
    fprintf(stderr, "   topList:"); for (uint32_t k = 0; k < t.result.topCount; ++k) fprintf(stderr, " %d", tl[k]); fprintf(stderr, "  (root %d)\n", t.root);
  }
  return 0;
}

// rtggx_build_as: every launch of the build on `s`, the tree arrays of all input sets, ONE wait.
int buildLbvh(rtggx_context* c, uint32_t slot, hipStream_t s) {
  MeshDev& m = c->mesh[slot];
  const uint32_t n = m.numTris;
  m.root = -1; m.depth = 0; m.stack4 = 0;
  freeBuildProducts(m);       // also the nodes / leaf triangles of every input set, and a rebuild in progress (the caller has synchronised)
  if (n == 0) return 0;
  // a mesh that deforms is built from its newest shape
  const float* verts = m.deforming ? m.vertsBuf[m.latestSet] : m.verts;
  BuildJob job; job.numTris = n; job.numVerts = m.numVerts;
  { const int r = allocTopo(job.topo, n); if (r) return r; }
  { const int r = allocScratch(job.s, n, m.numVerts, false); if (r) return r; }
  planBuildSteps(c, slot, job, verts, nullptr);
  for (auto& step : job.steps) step(s);
  RT_HIP(hipGetLastError());
  RT_HIP(hipStreamSynchronize(s));
  m.topo = job.topo; job.topo = BvhTopo{};
  freeScratch(job.s);
  { const int r = harvest(m, m.topo, slot); if (r) return r; }
  m.root = m.topo.root;
  m.builtCost = m.lastCost = m.topo.result.cost; m.costInFlight = false;
  // the tree arrays: one allocation aliased by all input sets while the mesh is static, one per set once it deforms
  const size_t nn = n > 1 ? n - 1 : 1, topCap = slot == 0 ? RT_TOP_SLOT0 : RT_TOP_SLOT1;
  const int sets = m.deforming ? RT_SETS : 1;
  for (int i = 0; i < sets; ++i) {
    RT_HIP(hipMalloc(&m.trisBuf[i], sizeof(BvhTri) * (size_t)n)); RT_HIP(hipMalloc(&m.nodesBuf[i], sizeof(BvhNode) * nn)); RT_HIP(hipMalloc(&m.nodes4Buf[i], sizeof(Bvh4Node) * nn));
    RT_HIP(hipMalloc(&m.topBuf[i], sizeof(Bvh4Node) * topCap));
  }
  for (int i = sets; i < RT_SETS; ++i) { m.trisBuf[i] = m.trisBuf[0]; m.nodesBuf[i] = m.nodesBuf[0]; m.nodes4Buf[i] = m.nodes4Buf[0]; m.topBuf[i] = m.topBuf[0]; }
  ++m.topoVersion;
  for (int i = 0; i < sets; ++i) {
    const float* v = m.deforming ? m.vertsBuf[i] : m.verts;
    if (m.deforming && m.topo.refittable) { const int r = refitLbvh(c, slot, (uint32_t)i, s); if (r) return r; }      // every set's tree from that set's own vertices
    else {
      hipLaunchKernelGGL(emitTris, dim3((n + 255) / 256), dim3(256), 0, s, (int)n, (const uint32_t*)m.topo.order, v, (const uint32_t*)m.indices, m.trisBuf[i]);
      emitTree(m, m.topo, (uint32_t)i, true, s);
    }
    m.topoVersionOfSet[i] = m.topoVersion;
  }
  for (int i = sets; i < RT_SETS; ++i) { m.topCountBuf[i] = m.topCountBuf[0]; m.topoVersionOfSet[i] = m.topoVersion; }
  RT_HIP(hipGetLastError());
  RT_HIP(hipStreamSynchronize(s));
  return 0;
}

// ---- a rebuild beside the frames (meshes that deform) -------------------------------------------------------------------------------
// A refit keeps the topology the last build chose for ANOTHER shape; when the tree's cost has drifted (capi.hip rtggx_refit_as) the mesh
// is built anew from the vertices of input set `set` -- copied first: the set moves on --, a few launches per frame on the stream the
// refits run on, behind the frame's refit.  When the last launch has ended (the host polls an event at the start of a frame) the new
// topology replaces the old one: the frame's refit, a moment later on the same stream, is the first to use it.  The buffers of the two
// topologies and the build's scratch memory are allocated once per mesh and swapped; nothing is freed, nothing waits.
// The second topology, the build's scratch memory and the event, allocated ONCE, when the mesh begins to deform (rtggx_refit_as's first
// call for it, which synchronises anyway): ~30 allocations that may serialise with work in flight have no place in the frame loop.  A
// failure leaves no half-made job behind (a later startRebuild would plan kernels over null pointers).
int prepareRebuild(rtggx_context* c, uint32_t slot) {
  MeshDev& m = c->mesh[slot];
  if (m.job || !m.topo.refittable || m.numTris < 2) return 0;
  BuildJob* job = new BuildJob();
  job->numTris = m.numTris; job->numVerts = m.numVerts;
  int r = allocTopo(job->topo, m.numTris);
  if (!r) r = allocScratch(job->s, m.numTris, m.numVerts, true);
  if (!r && hipEventCreateWithFlags(&job->done, hipEventDisableTiming) != hipSuccess) { setError("prepareRebuild: hipEventCreate failed"); r = -2; }
  if (r) { if (job->done) hipEventDestroy(job->done); freeTopo(job->topo); freeScratch(job->s); delete job; return r; }
  m.job = job;
  return 0;
}
int startRebuild(rtggx_context* c, uint32_t slot, uint32_t set) {
  MeshDev& m = c->mesh[slot];
  if (!m.deforming || !m.topo.refittable || m.numTris < 2) return 0;
  if (!m.job) { const int r = prepareRebuild(c, slot); if (r) return r; }      // (a mesh that was built again since it began to deform: buildLbvh frees the job)
  BuildJob& job = *m.job;
  if (job.active) return 0;
  planBuildSteps(c, slot, job, nullptr, m.vertsBuf[set]);
  job.active = true;
  return 1;      // started
}
int continueRebuild(rtggx_context* c, uint32_t slot, hipStream_t s, uint32_t maxSteps, bool* swapped) {
  MeshDev& m = c->mesh[slot];
  *swapped = false;
  if (!m.job || !m.job->active) return 0;
  BuildJob& job = *m.job;
  if (job.allIssued) {
    if (maxSteps != 0u || hipEventQuery(job.done) != hipSuccess) return 0;      // (asked to issue: nothing left to) / still running
    job.active = false;
    { const int r = harvest(m, job.topo, slot); if (r) return r; }
    std::swap(m.topo, job.topo);
    m.root = m.topo.root; m.builtCost = m.lastCost = m.topo.result.cost; m.costInFlight = false;
    ++m.topoVersion; ++m.rebuilds;
    *swapped = true;
    return 0;
  }
  if (maxSteps == 0u) return 0;
  for (uint32_t k = 0; k < maxSteps && job.next < job.steps.size(); ++k) job.steps[job.next++](s);
  RT_HIP(hipGetLastError());
  if (job.next == job.steps.size()) { RT_HIP(hipEventRecord(job.done, s)); job.allIssued = true; }
  return 0;
}
void abandonRebuild(rtggx_context* c, uint32_t slot) {
  MeshDev& m = c->mesh[slot];
  if (m.job && m.job->active) { hipDeviceSynchronize(); m.job->active = false; }
}


static int launchTreeCost(MeshDev& m, hipStream_t s) {
  if (!m.dCost) { RT_HIP(hipMalloc(&m.dCost, 4)); RT_HIP(hipHostMalloc(&m.hCost, 4)); *m.hCost = 0.0f; RT_HIP(hipEventCreateWithFlags(&m.evCost, hipEventDisableTiming)); }
  RT_HIP(hipMemsetAsync(m.dCost, 0, 4, s));
  const int numNodes = (int)m.numTris - 1;
  hipLaunchKernelGGL(treeCostKernel, dim3(std::min((numNodes + 255) / 256, 48)), dim3(256), 0, s, numNodes, (const float*)m.topo.nodeBox, m.dCost);
  return 0;
}

// New boxes for the existing topology from the vertex buffer of input set `set`, into that set's leaf triangles and nodes: the fat
// triangles, the leaf triangles + per-primitive boxes, the node boxes treelet level by treelet level (refitTreelets), the
// 64-byte binary nodes (for the oracle / tests) and their 4-wide collapse (for the trace kernel), every 4th time the tree's cost.
// No host round trip; everything on stream `s` (stream R, beside whatever the other streams are doing).
int refitLbvh(rtggx_context* c, uint32_t slot, uint32_t set, hipStream_t s) {
  MeshDev& m = c->mesh[slot];
  const BvhTopo& t = m.topo;
  const uint32_t n = m.numTris;
  if (n == 0 || !m.trisBuf[set]) return 0;
  const uint32_t nb = (n + 255) / 256;
  if (!t.triBox || (n > 1 && !t.refittable)) { setError("rtggx_refit_as: mesh %u has no PLOC build to refit (RTGGX_BVH_RADIX_TREE builds cannot be refitted)", slot); return -1; }
  { const int r = buildFatTris(c, slot, set, s); if (r) return r; }
  hipLaunchKernelGGL(refitTris, dim3(nb), dim3(256), 0, s, (int)n, (const uint32_t*)t.order, (const float*)m.vertsBuf[set], (const uint32_t*)m.indices, m.trisBuf[set], t.triBox);
  if (n > 1) {
    uint32_t first = 0;
    for (int l = 0; l < RT_TREELET_LEVELS; ++l) {      // level after level
      const uint32_t k = t.result.treelets[l];
      if (k) hipLaunchKernelGGL(refitTreelets, dim3(k), dim3(256), 0, s, (const RefitTreelet*)t.dTreelets + first, (const int4*)t.dRefitItems, (const uint32_t*)t.dRefitRounds,
                                (const uint32_t*)t.order, (const float*)t.triBox, t.nodeBox);
      first += k;
    }
  }
  emitTree(m, t, set, m.topoVersionOfSet[set] != m.topoVersion, s);
  m.topoVersionOfSet[set] = m.topoVersion;
  if (n > 1 && !m.costInFlight && (m.refits & 3u) == 0u) {
    { const int r = launchTreeCost(m, s); if (r) return r; }
    RT_HIP(hipMemcpyAsync(m.hCost, m.dCost, 4, hipMemcpyDeviceToHost, s));
    RT_HIP(hipEventRecord(m.evCost, s));
    m.costInFlight = true;
  }
  ++m.refits;
  RT_HIP(hipGetLastError());
  return 0;
}

}  // namespace rt
