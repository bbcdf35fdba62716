// Bottom-level acceleration structures: the gfx950 replacement of the driver-side BLAS build the
// reference requests in RayTracer::buildAccelerationStructures / BuildAccelerationStructures
// (RayTracedGGX/Content/RayTracer.cpp:676-716, 158-233; PREFER_FAST_TRACE, one triangle geometry
// per mesh, R32G32B32_FLOAT positions at stride 24, 32-bit indices).
//
// LBVH (Karras 2012): 30-bit Morton codes of triangle-box centres -> stable LSD radix sort
// (4 x 8 bits, wave64 ballot ranking) -> radix-tree hierarchy, one lane per internal node ->
// bottom-up box fit with one arrival counter per node -> 64-byte nodes that carry both child
// boxes, and 64-byte leaf triangles in Morton order.  The build runs on the context's build stream with host round trips
// (PLOC rounds) and is off the per-frame path; what IS on it, for meshes that change shape, is refitLbvh at the end of
// this file: new leaf triangles and boxes for the existing topology, five kernels on stream B, no host involvement.
#include <algorithm>
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

__global__ void mortonKernel(const float* __restrict__ verts, const uint32_t* __restrict__ idx, uint32_t n,
                             float3 bmin, float3 invExt, uint32_t* __restrict__ codes, uint32_t* __restrict__ order,
                             float* __restrict__ triBox) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
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
// exclusive scan of `count` values by one workgroup
__global__ void __launch_bounds__(1024) scanExclusive(uint32_t* __restrict__ data, uint32_t count) {
  __shared__ uint32_t partial[1024];
  const uint32_t per = (count + 1023) / 1024;
  const uint32_t b = threadIdx.x * per, e = min(b + per, count);
  uint32_t s = 0;
  for (uint32_t i = b; i < e; ++i) s += data[i];
  partial[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) { uint32_t run = 0; for (int i = 0; i < 1024; ++i) { const uint32_t v = partial[i]; partial[i] = run; run += v; } }
  __syncthreads();
  uint32_t run = partial[threadIdx.x];
  for (uint32_t i = b; i < e; ++i) { const uint32_t v = data[i]; data[i] = run; run += v; }
}
__global__ void __launch_bounds__(256) radixScatter(const uint32_t* __restrict__ keysIn, const uint32_t* __restrict__ valsIn, uint32_t n, int shift,
                                                    const uint32_t* __restrict__ hist, uint32_t numBlocks,
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
    uint32_t off = hist[digit * numBlocks + blockIdx.x] + rankInWave;
    for (uint32_t w = 0; w < wave; ++w) off += waveCount[w][digit];
    keysOut[off] = key; valsOut[off] = valsIn[i];
  }
}

// ---- Karras hierarchy --------------------------------------------------------------------------------
RT_DEV int deltaLcp(const uint32_t* __restrict__ codes, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  const uint32_t a = codes[i], b = codes[j];
  if (a == b) return 32 + __clz((uint32_t)i ^ (uint32_t)j);
  return __clz(a ^ b);
}
__global__ void hierarchyKernel(const uint32_t* __restrict__ codes, int n, int32_t* __restrict__ left, int32_t* __restrict__ right,
                                int32_t* __restrict__ nodeParent, int32_t* __restrict__ leafParent) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1) return;
  const int d = (deltaLcp(codes, n, i, i + 1) - deltaLcp(codes, n, i, i - 1)) >= 0 ? 1 : -1;
  const int dmin = deltaLcp(codes, n, i, i - d);
  int lmax = 2;
  while (deltaLcp(codes, n, i, i + lmax * d) > dmin) lmax *= 2;
  int l = 0;
  for (int t = lmax / 2; t >= 1; t /= 2) if (deltaLcp(codes, n, i, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  const int dnode = deltaLcp(codes, n, i, j);
  int s = 0, t = l;
  do { t = (t + 1) / 2; if (deltaLcp(codes, n, i, i + (s + t) * d) > dnode) s += t; } while (t > 1);
  const int gamma = i + s * d + min(d, 0);
  const int lo = min(i, j), hi = max(i, j);
  if (lo == gamma) { left[i] = ~gamma; leafParent[gamma] = i; } else { left[i] = gamma; nodeParent[gamma] = i; }
  if (hi == gamma + 1) { right[i] = ~(gamma + 1); leafParent[gamma + 1] = i; } else { right[i] = gamma + 1; nodeParent[gamma + 1] = i; }
  if (i == 0) nodeParent[0] = -1;
}

RT_DEV float ldAgent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
RT_DEV void stAgent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// One lane per leaf climbs towards the root; the second arrival at a node merges the child boxes.
__global__ void fitKernel(int n, const uint32_t* __restrict__ order, const float* __restrict__ triBox, const int32_t* __restrict__ left,
                          const int32_t* __restrict__ right, const int32_t* __restrict__ nodeParent, const int32_t* __restrict__ leafParent,
                          float* nodeBox, uint32_t* arrive) {
  const int leaf = blockIdx.x * blockDim.x + threadIdx.x;
  if (leaf >= n) return;
  int cur = leafParent[leaf];
  while (cur >= 0) {
    __threadfence();
    const uint32_t old = atomicAdd(&arrive[cur], 1u);
    if (old == 0) return;
    __threadfence();
    float b[2][6];
    const int32_t ch[2] = {left[cur], right[cur]};
    for (int s = 0; s < 2; ++s) {
      if (ch[s] < 0) { const uint32_t prim = order[~ch[s]]; for (int k = 0; k < 6; ++k) b[s][k] = triBox[6 * (size_t)prim + k]; }
      else for (int k = 0; k < 6; ++k) b[s][k] = ldAgent(&nodeBox[6 * (size_t)ch[s] + k]);
    }
    for (int k = 0; k < 3; ++k) { stAgent(&nodeBox[6 * (size_t)cur + k], fminf(b[0][k], b[1][k])); stAgent(&nodeBox[6 * (size_t)cur + 3 + k], fmaxf(b[0][3 + k], b[1][3 + k])); }
    cur = nodeParent[cur];
  }
}

// ---- PLOC: parallel locally-ordered clustering (Meister & Bittner 2018) ------------------------------------
// Agglomerative build over the Morton-ordered triangles: every cluster looks RT_PLOC_RADIUS positions to either side
// for the partner that gives the smallest merged box; mutual choices merge into a new node; the survivors are
// compacted (order kept) and the step repeats until one cluster is left.  Trees come out close to a SAH sweep
// build -- on the bunny a third fewer traversal steps per ray than the Karras radix tree (profiles/r01_d) -- and the
// build is off the frame path.  Everything is deterministic: ties go to the lower position, node indices come from
// prefix sums.
#define RT_PLOC_RADIUS 16
RT_DEV float mergedArea(const float* a, const float* b) {
  const float ex = fmaxf(a[3], b[3]) - fminf(a[0], b[0]), ey = fmaxf(a[4], b[4]) - fminf(a[1], b[1]), ez = fmaxf(a[5], b[5]) - fminf(a[2], b[2]);
  return (ex * ey + ey * ez) + ez * ex;
}
__global__ void plocInit(int n, const uint32_t* __restrict__ order, const float* __restrict__ triBox, int32_t* __restrict__ clRef, float* __restrict__ clBox) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  clRef[s] = ~s;
  for (int k = 0; k < 6; ++k) clBox[6 * (size_t)s + k] = triBox[6 * (size_t)order[s] + k];
}
__global__ void plocNearest(int m, int radius, const float* __restrict__ clBox, int32_t* __restrict__ nn) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  float mine[6];
  for (int k = 0; k < 6; ++k) mine[k] = clBox[6 * (size_t)i + k];
  float best = __builtin_inff(); int bj = -1;
  const int lo = max(i - radius, 0), hi = min(i + radius, m - 1);
  for (int j = lo; j <= hi; ++j) {
    if (j == i) continue;
    const float d = mergedArea(mine, clBox + 6 * (size_t)j);
    if (d < best) { best = d; bj = j; }
  }
  nn[i] = bj;
}
__global__ void plocFlags(int m, const int32_t* __restrict__ nn, uint32_t* __restrict__ keep, uint32_t* __restrict__ merge) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const int j = nn[i];
  const bool mutual = j >= 0 && nn[j] == i;
  merge[i] = mutual && i < j ? 1u : 0u;        // the lower position carries the new node
  keep[i] = mutual && i > j ? 0u : 1u;         // the higher one disappears
}
// keepPos / mergePos: exclusive prefix sums of the flags (the flags themselves are recovered from nn)
__global__ void plocScatter(int m, int nodeBase, const int32_t* __restrict__ nn, const uint32_t* __restrict__ keepPos, const uint32_t* __restrict__ mergePos,
                            const int32_t* __restrict__ clRef, const float* __restrict__ clBox, int32_t* __restrict__ clRefOut, float* __restrict__ clBoxOut,
                            int32_t* __restrict__ left, int32_t* __restrict__ right, int32_t* __restrict__ nodeParent, int32_t* __restrict__ leafParent,
                            float* __restrict__ nodeBox) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const int j = nn[i];
  const bool mutual = j >= 0 && nn[j] == i;
  if (mutual && i > j) return;
  const uint32_t p = keepPos[i];
  if (mutual) {
    const int node = nodeBase + (int)mergePos[i];
    const int32_t L = clRef[i], R = clRef[j];
    left[node] = L; right[node] = R;
    if (L < 0) leafParent[~L] = node; else nodeParent[L] = node;
    if (R < 0) leafParent[~R] = node; else nodeParent[R] = node;
    for (int k = 0; k < 3; ++k) {
      const float mn = fminf(clBox[6 * (size_t)i + k], clBox[6 * (size_t)j + k]), mx = fmaxf(clBox[6 * (size_t)i + 3 + k], clBox[6 * (size_t)j + 3 + k]);
      nodeBox[6 * (size_t)node + k] = mn; nodeBox[6 * (size_t)node + 3 + k] = mx;
      clBoxOut[6 * (size_t)p + k] = mn; clBoxOut[6 * (size_t)p + 3 + k] = mx;
    }
    clRefOut[p] = node;
  } else {
    clRefOut[p] = clRef[i];
    for (int k = 0; k < 6; ++k) clBoxOut[6 * (size_t)p + k] = clBox[6 * (size_t)i + k];
  }
}

// Leaf depth = number of ancestors = the most stack entries a traversal reaching that leaf can hold.
__global__ void depthKernel(int n, const int32_t* __restrict__ nodeParent, const int32_t* __restrict__ leafParent, uint32_t* __restrict__ maxDepth) {
  const int leaf = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t d = 0;
  if (leaf < n) for (int cur = leafParent[leaf]; cur >= 0; cur = nodeParent[cur]) ++d;
  for (int o = 32; o > 0; o >>= 1) d = max(d, (uint32_t)__shfl_down((int)d, o));
  if ((threadIdx.x & 63) == 0 && d) atomicMax(maxDepth, d);
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
// 4-wide collapse: every internal node of even depth becomes a Bvh4Node whose entries are its grandchildren (or
// its children where those are leaves).  Boxes are the binary tree's own child boxes, so the set of triangles a
// ray reaches can only grow relative to the binary traversal (one box test per two levels is skipped).
__global__ void emitNodes4(int n, const uint32_t* __restrict__ order, const float* __restrict__ triBox, const int32_t* __restrict__ left,
                           const int32_t* __restrict__ right, const int32_t* __restrict__ nodeParent, const float* __restrict__ nodeBox,
                           Bvh4Node* __restrict__ nodes4) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1) return;
  int depth = 0;
  for (int p = nodeParent[i]; p >= 0; p = nodeParent[p]) ++depth;
  if (depth & 1) return;
  int32_t refs[4]; int cnt = 0;
  const int32_t ch[2] = {left[i], right[i]};
  for (int s = 0; s < 2; ++s) {
    if (ch[s] < 0) refs[cnt++] = ch[s];
    else { refs[cnt++] = left[ch[s]]; refs[cnt++] = right[ch[s]]; }
  }
  Bvh4Node nd;
  for (int k = 0; k < 4; ++k) {
    if (k < cnt) {
      const float* b = refs[k] < 0 ? &triBox[6 * (size_t)order[~refs[k]]] : &nodeBox[6 * (size_t)refs[k]];
      nd.minx[k] = b[0]; nd.miny[k] = b[1]; nd.minz[k] = b[2]; nd.maxx[k] = b[3]; nd.maxy[k] = b[4]; nd.maxz[k] = b[5];
      nd.ref[k] = refs[k];
    } else {
      nd.minx[k] = nd.miny[k] = nd.minz[k] = __builtin_inff(); nd.maxx[k] = nd.maxy[k] = nd.maxz[k] = -__builtin_inff();
      nd.ref[k] = RT_BVH4_EMPTY;
    }
    nd.pad[k] = 0;
  }
  nd.pad[0] = depth >> 1;      // level in the 4-wide tree (read by -DRT_TRACE_STATS builds only)
  nodes4[i] = nd;
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

// ---- refit (rtggx_refit_as): same topology, new vertex positions ---------------------------------------------------------
// Leaf slot s keeps its primitive (tris[s].prim): rewrite its three vertices and the primitive's box.
__global__ void refitTris(int n, const float* __restrict__ verts, const uint32_t* __restrict__ idx, BvhTri* __restrict__ tris, float* __restrict__ triBox) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const uint32_t prim = tris[s].prim;
  float mn[3], mx[3];
  BvhTri t = tris[s];
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
__global__ void treeCostKernel(int numNodes, const float* __restrict__ nodeBox, float* __restrict__ cost) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float a = 0.0f;
  if (i < numNodes) {
    const float* b = nodeBox + 6 * (size_t)i;
    const float ex = b[3] - b[0], ey = b[4] - b[1], ez = b[5] - b[2];
    a = (ex * ey + ey * ez) + ez * ex;
  }
  for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o);
  if ((threadIdx.x & 63) == 0 && a != 0.0f) atomicAdd(cost, a);
}

// Bottom-up box refit without global synchronisation: TREELETS.  At build time the host cuts the tree into subtrees of at most
// RT_TREELET_NODES internal nodes (planRefit below); one workgroup refits one treelet, round by round of the PLOC build (a node's
// children belong to earlier rounds), with the treelet's boxes in LDS and a workgroup barrier between rounds.  What is left above
// the treelet roots is cut the same way again, until one treelet holds the root: two launches for the bunny and the dragon.
// A child reference of an item is: >= 0 the position of another item of the same treelet (its box is in LDS);
// 0xC0000000 | s leaf slot s (the primitive's box); 0x80000000 | n node n of an earlier level (its box is final in nodeBox).
#define RT_TREELET_NODES 1024
struct RefitTreelet { uint32_t itemBegin, roundBegin, numRounds, pad; };
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
void freeBuildProducts(MeshDev& m) {
  hipFree(m.order); hipFree(m.left); hipFree(m.right); hipFree(m.nodeParent); hipFree(m.leafParent); hipFree(m.nodeBox); hipFree(m.triBox); hipFree(m.dTreelets); hipFree(m.dRefitItems); hipFree(m.dRefitRounds);
  m.order = nullptr; m.left = m.right = m.nodeParent = m.leafParent = nullptr; m.nodeBox = m.triBox = nullptr; m.dTreelets = nullptr; m.dRefitItems = nullptr; m.dRefitRounds = nullptr;
  m.roundBase.clear(); m.refitLevels.clear();
  freeAliased(m.nodesBuf); freeAliased(m.nodes4Buf); freeAliased(m.trisBuf); freeAliased(m.topBuf);
  hipFree(m.topList); hipFree(m.topRank); m.topList = m.topRank = nullptr; m.topCount = 0;
  m.nodes = nullptr; m.nodes4 = nullptr; m.tris = nullptr; m.top = nullptr;
}

static int launchTreeCost(MeshDev& m, hipStream_t s) {
  if (!m.dCost) { RT_HIP(hipMalloc(&m.dCost, 4)); RT_HIP(hipHostMalloc(&m.hCost, 4)); *m.hCost = 0.0f; RT_HIP(hipEventCreateWithFlags(&m.evCost, hipEventDisableTiming)); }
  RT_HIP(hipMemsetAsync(m.dCost, 0, 4, s));
  const int numNodes = (int)m.numTris - 1;
  hipLaunchKernelGGL(treeCostKernel, dim3((numNodes + 255) / 256), dim3(256), 0, s, numNodes, m.nodeBox, m.dCost);
  return 0;
}

// New boxes for the existing tree from the vertex buffer of input set `set`, into that set's leaf triangles and nodes: the leaf
// triangles, the node boxes treelet level by treelet level (refitTreelets), the
// 64-byte binary nodes (for the oracle / tests) and their 4-wide collapse (for the trace kernel), every 8th time the tree's cost.
// No host round trip; everything on stream `s` (stream R, beside whatever the other streams are doing).
int refitLbvh(rtggx_context* c, uint32_t slot, uint32_t set, hipStream_t s) {
  MeshDev& m = c->mesh[slot];
  const uint32_t n = m.numTris;
  if (n == 0 || !m.trisBuf[set]) return 0;
  const uint32_t nb = (n + 255) / 256;
  if (!m.triBox || (n > 1 && m.refitLevels.empty())) { setError("rtggx_refit_as: mesh %u has no PLOC build to refit (RTGGX_BVH_RADIX_TREE builds cannot be refitted)", slot); return -1; }
  { const int r = buildFatTris(c, slot, set, s); if (r) return r; }
  hipLaunchKernelGGL(refitTris, dim3(nb), dim3(256), 0, s, (int)n, (const float*)m.vertsBuf[set], (const uint32_t*)m.indices, m.trisBuf[set], m.triBox);
  if (n > 1) {
    for (const auto& lv : m.refitLevels)      // (first treelet, count): level after level
      hipLaunchKernelGGL(refitTreelets, dim3(lv.second), dim3(256), 0, s, (const RefitTreelet*)m.dTreelets + lv.first, (const int4*)m.dRefitItems, (const uint32_t*)m.dRefitRounds,
                         (const uint32_t*)m.order, (const float*)m.triBox, m.nodeBox);
    hipLaunchKernelGGL(emitNodes, dim3(nb), dim3(256), 0, s, (int)n, (const uint32_t*)m.order, (const float*)m.triBox, (const int32_t*)m.left, (const int32_t*)m.right, (const float*)m.nodeBox, m.nodesBuf[set]);
    hipLaunchKernelGGL(emitNodes4, dim3(nb), dim3(256), 0, s, (int)n, (const uint32_t*)m.order, (const float*)m.triBox, (const int32_t*)m.left, (const int32_t*)m.right,
                       (const int32_t*)m.nodeParent, (const float*)m.nodeBox, m.nodes4Buf[set]);
    if (m.topCount) hipLaunchKernelGGL(emitTop, dim3((m.topCount + 63) / 64), dim3(64), 0, s, (int)m.topCount, (const int32_t*)m.topList, (const int32_t*)m.topRank, (const Bvh4Node*)m.nodes4Buf[set], m.topBuf[set]);
    if (!m.costInFlight && (m.refits & 7u) == 0u) {
      { const int r = launchTreeCost(m, s); if (r) return r; }
      RT_HIP(hipMemcpyAsync(m.hCost, m.dCost, 4, hipMemcpyDeviceToHost, s));
      RT_HIP(hipEventRecord(m.evCost, s));
      m.costInFlight = true;
    }
  }
  ++m.refits;
  RT_HIP(hipGetLastError());
  return 0;
}

// The first `capacity` 4-wide nodes in breadth-first order (4-wide nodes = the binary nodes of even depth, rtggx_device.h): their
// list, every node's rank in it, and the table emitTop makes of them.  Host side, at build time.
static int planTop(MeshDev& m, uint32_t capacity, const std::vector<int32_t>& left, const std::vector<int32_t>& right, hipStream_t s) {
  const int numNodes = (int)m.numTris - 1;
  std::vector<int32_t> list, rank(numNodes, -1);
  if (m.root >= 0) list.push_back(m.root);
  for (size_t head = 0; head < list.size() && list.size() < capacity; ++head) {      // breadth first: `list` is the queue
    const int32_t v = list[head];
    const int32_t ch[2] = {left[v], right[v]};
    for (int side = 0; side < 2; ++side) {
      if (ch[side] < 0) continue;
      const int32_t g[2] = {left[ch[side]], right[ch[side]]};
      for (int k = 0; k < 2; ++k) if (g[k] >= 0 && list.size() < capacity) list.push_back(g[k]);
    }
  }
  for (size_t k = 0; k < list.size(); ++k) rank[list[k]] = (int32_t)k;
  m.topCount = (uint32_t)list.size();
  if (m.topCount == 0) return 0;
  RT_HIP(hipMalloc(&m.topList, 4 * list.size())); RT_HIP(hipMalloc(&m.topRank, 4 * (size_t)numNodes)); RT_HIP(hipMalloc(&m.topBuf[0], sizeof(Bvh4Node) * list.size()));
  for (int i = 1; i < RT_SETS; ++i) m.topBuf[i] = m.topBuf[0];
  m.top = m.topBuf[0];
  RT_HIP(hipMemcpyAsync(m.topList, list.data(), 4 * list.size(), hipMemcpyHostToDevice, s)); RT_HIP(hipMemcpyAsync(m.topRank, rank.data(), 4 * (size_t)numNodes, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(emitTop, dim3((m.topCount + 63) / 64), dim3(64), 0, s, (int)m.topCount, (const int32_t*)m.topList, (const int32_t*)m.topRank, (const Bvh4Node*)m.nodes4, m.topBuf[0]);
  RT_HIP(hipStreamSynchronize(s));      // list / rank are host vectors
  return 0;
}

// The refit schedule of a freshly built PLOC tree (host side; the build is synchronous anyway).  left / right: the children of the
// numNodes internal nodes as the build left them (>= 0 node, < 0 ~leaf slot); PLOC hands out node indices round by round, so a
// child's index is smaller than its parent's and roundBase tells a node's round.
static int planRefit(MeshDev& m, const std::vector<int32_t>& left, const std::vector<int32_t>& right, hipStream_t s) {
  const int numNodes = (int)m.numTris - 1;
  std::vector<int32_t> parent(numNodes, -1), roundOf(numNodes, 0);
  for (int i = 0; i < numNodes; ++i) { if (left[i] >= 0) parent[left[i]] = i; if (right[i] >= 0) parent[right[i]] = i; }
  for (size_t k = 0; k + 1 < m.roundBase.size(); ++k) for (uint32_t i = m.roundBase[k]; i < m.roundBase[k + 1]; ++i) roundOf[i] = (int32_t)k;
  std::vector<uint8_t> pending(numNodes, 1);      // nodes no level has taken yet
  std::vector<uint32_t> cnt(numNodes);
  std::vector<int32_t> local(numNodes, -1);
  std::vector<RefitTreelet> treelets; std::vector<int4> items; std::vector<uint32_t> rounds;
  m.refitLevels.clear();
  int remaining = numNodes;
  while (remaining > 0) {
    // pending nodes in each subtree (children come first in index order)
    for (int i = 0; i < numNodes; ++i) {
      cnt[i] = pending[i] ? 1u : 0u;
      if (pending[i]) { if (left[i] >= 0) cnt[i] += cnt[left[i]]; if (right[i] >= 0) cnt[i] += cnt[right[i]]; }
    }
    const uint32_t firstTreelet = (uint32_t)treelets.size();
    for (int root = 0; root < numNodes; ++root) {
      if (!pending[root] || cnt[root] > RT_TREELET_NODES) continue;
      if (parent[root] >= 0 && cnt[parent[root]] <= RT_TREELET_NODES) continue;      // an inner node of somebody else's treelet
      // the pending nodes below `root`, by PLOC round (then by index: deterministic)
      std::vector<int32_t> nodes, stack{root};
      while (!stack.empty()) {
        const int32_t v = stack.back(); stack.pop_back();
        nodes.push_back(v);
        if (left[v] >= 0 && pending[left[v]]) stack.push_back(left[v]);
        if (right[v] >= 0 && pending[right[v]]) stack.push_back(right[v]);
      }
      std::sort(nodes.begin(), nodes.end(), [&](int32_t a, int32_t b) { return roundOf[a] != roundOf[b] ? roundOf[a] < roundOf[b] : a < b; });
      RefitTreelet tl{(uint32_t)items.size(), (uint32_t)rounds.size(), 0u, 0u};
      for (size_t k = 0; k < nodes.size(); ++k) local[nodes[k]] = (int32_t)k;
      for (size_t k = 0; k < nodes.size(); ++k) {
        if (k == 0 || roundOf[nodes[k]] != roundOf[nodes[k - 1]]) { rounds.push_back((uint32_t)k); ++tl.numRounds; }
        const int32_t v = nodes[k];
        const auto ref = [&](int32_t ch) -> int32_t {
          if (ch < 0) return (int32_t)(0xC0000000u | (uint32_t)~ch);                        // leaf slot
          if (pending[ch]) return local[ch];                                                // same treelet (its subtree is closed)
          return (int32_t)(0x80000000u | (uint32_t)ch);                                     // finished by an earlier level
        };
        items.push_back(make_int4(v, ref(left[v]), ref(right[v]), 0));
      }
      rounds.push_back((uint32_t)nodes.size());
      treelets.push_back(tl);
      for (int32_t v : nodes) { pending[v] = 2; }      // taken by this level (still "pending" for the scan of this level's other roots)
      remaining -= (int)nodes.size();
    }
    for (int i = 0; i < numNodes; ++i) if (pending[i] == 2) pending[i] = 0;
    if (treelets.size() == firstTreelet) { setError("planRefit: no progress with %d nodes left", remaining); return -3; }
    m.refitLevels.push_back({firstTreelet, (uint32_t)treelets.size() - firstTreelet});
  }
  RT_HIP(hipMalloc(&m.dTreelets, sizeof(RefitTreelet) * treelets.size()));
  RT_HIP(hipMalloc(&m.dRefitItems, sizeof(int4) * items.size()));
  RT_HIP(hipMalloc(&m.dRefitRounds, 4 * rounds.size()));
  RT_HIP(hipMemcpyAsync(m.dTreelets, treelets.data(), sizeof(RefitTreelet) * treelets.size(), hipMemcpyHostToDevice, s));
  RT_HIP(hipMemcpyAsync(m.dRefitItems, items.data(), sizeof(int4) * items.size(), hipMemcpyHostToDevice, s));
  RT_HIP(hipMemcpyAsync(m.dRefitRounds, rounds.data(), 4 * rounds.size(), hipMemcpyHostToDevice, s));
  RT_HIP(hipStreamSynchronize(s));      // the host vectors go out of scope
  return 0;
}

int buildLbvh(rtggx_context* c, uint32_t slot, hipStream_t s) {
  MeshDev& m = c->mesh[slot];
  const uint32_t n = m.numTris;
  m.root = -1; m.depth = 0;
  freeBuildProducts(m);       // also the nodes / leaf triangles of every input set
  if (n == 0) return 0;
  RT_HIP(hipMalloc(&m.tris, sizeof(BvhTri) * (size_t)n));
  RT_HIP(hipMalloc(&m.nodes, sizeof(BvhNode) * (size_t)(n > 1 ? n - 1 : 1)));
  RT_HIP(hipMalloc(&m.nodes4, sizeof(Bvh4Node) * (size_t)(n > 1 ? n - 1 : 1)));
  RT_HIP(hipMemsetAsync(m.nodes4, 0, sizeof(Bvh4Node) * (size_t)(n > 1 ? n - 1 : 1), s));
  for (int i = 0; i < RT_SETS; ++i) { m.nodesBuf[i] = m.nodes; m.nodes4Buf[i] = m.nodes4; m.trisBuf[i] = m.tris; }     // one allocation for all input sets until the mesh deforms

  const float* mn = m.bmin; const float* mx = m.bmax;   // vertex bounds recorded by rtggx_set_mesh
  float3 bmin = make_float3(mn[0], mn[1], mn[2]);
  float3 invExt = make_float3(mx[0] > mn[0] ? 1.0f / (mx[0] - mn[0]) : 0.0f, mx[1] > mn[1] ? 1.0f / (mx[1] - mn[1]) : 0.0f, mx[2] > mn[2] ? 1.0f / (mx[2] - mn[2]) : 0.0f);

  const uint32_t nb = (n + 255) / 256;
  uint32_t *codes[2], *order[2], *hist; float *triBox, *nodeBox; int32_t *left, *right, *nodeParent, *leafParent; uint32_t* arrive;
  RT_HIP(hipMalloc(&codes[0], 4 * (size_t)n)); RT_HIP(hipMalloc(&codes[1], 4 * (size_t)n));
  RT_HIP(hipMalloc(&order[0], 4 * (size_t)n)); RT_HIP(hipMalloc(&order[1], 4 * (size_t)n));
  RT_HIP(hipMalloc(&hist, 4 * (size_t)256 * nb));
  RT_HIP(hipMalloc(&triBox, 4 * 6 * (size_t)n)); RT_HIP(hipMalloc(&nodeBox, 4 * 6 * (size_t)n));
  RT_HIP(hipMalloc(&left, 4 * (size_t)n)); RT_HIP(hipMalloc(&right, 4 * (size_t)n));
  RT_HIP(hipMalloc(&nodeParent, 4 * (size_t)n)); RT_HIP(hipMalloc(&leafParent, 4 * (size_t)n));
  RT_HIP(hipMalloc(&arrive, 4 * ((size_t)n + 1)));            // + 1: the depthKernel result
  RT_HIP(hipMemsetAsync(arrive, 0, 4 * ((size_t)n + 1), s));

  hipLaunchKernelGGL(mortonKernel, dim3(nb), dim3(256), 0, s, m.verts, m.indices, n, bmin, invExt, codes[0], order[0], triBox);
  int cur = 0;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = pass * 8;
    hipLaunchKernelGGL(radixHist, dim3(nb), dim3(256), 0, s, codes[cur], n, shift, hist, nb);
    hipLaunchKernelGGL(scanExclusive, dim3(1), dim3(1024), 0, s, hist, 256u * nb);
    hipLaunchKernelGGL(radixScatter, dim3(nb), dim3(256), 0, s, codes[cur], order[cur], n, shift, hist, nb, codes[cur ^ 1], order[cur ^ 1]);
    cur ^= 1;
  }
  hipLaunchKernelGGL(emitTris, dim3(nb), dim3(256), 0, s, (int)n, order[cur], m.verts, m.indices, m.tris);
  if (n == 1) m.root = ~0;
  else {
    static const bool radixTree = getenv("RTGGX_BVH_RADIX_TREE") != nullptr;     // A/B switch: the Karras tree this build started with
    int32_t root = 0;
    if (radixTree) {
      hipLaunchKernelGGL(hierarchyKernel, dim3(nb), dim3(256), 0, s, codes[cur], (int)n, left, right, nodeParent, leafParent);
      hipLaunchKernelGGL(fitKernel, dim3(nb), dim3(256), 0, s, (int)n, order[cur], triBox, left, right, nodeParent, leafParent, nodeBox, arrive);
    } else {
      int32_t *clRef[2], *nn; float* clBox[2]; uint32_t *keepPos, *mergePos;
      RT_HIP(hipMalloc(&clRef[0], 4 * (size_t)n)); RT_HIP(hipMalloc(&clRef[1], 4 * (size_t)n)); RT_HIP(hipMalloc(&nn, 4 * (size_t)n));
      RT_HIP(hipMalloc(&clBox[0], 24 * (size_t)n)); RT_HIP(hipMalloc(&clBox[1], 24 * (size_t)n));
      RT_HIP(hipMalloc(&keepPos, 4 * (size_t)n)); RT_HIP(hipMalloc(&mergePos, 4 * (size_t)n));
      hipLaunchKernelGGL(plocInit, dim3(nb), dim3(256), 0, s, (int)n, order[cur], triBox, clRef[0], clBox[0]);
      int m = (int)n, nodeBase = 0, a = 0;
      std::vector<uint32_t> roundBase;
      const int radius = getenv("RTGGX_PLOC_RADIUS") ? atoi(getenv("RTGGX_PLOC_RADIUS")) : RT_PLOC_RADIUS;
      while (m > 1) {
        const dim3 g((m + 255) / 256);
        hipLaunchKernelGGL(plocNearest, g, dim3(256), 0, s, m, radius, clBox[a], nn);
        hipLaunchKernelGGL(plocFlags, g, dim3(256), 0, s, m, nn, keepPos, mergePos);
        uint32_t lastFlags[2], lastPos[2];     // totals = last exclusive prefix + last flag
        RT_HIP(hipMemcpyAsync(&lastFlags[0], keepPos + (m - 1), 4, hipMemcpyDeviceToHost, s)); RT_HIP(hipMemcpyAsync(&lastFlags[1], mergePos + (m - 1), 4, hipMemcpyDeviceToHost, s));
        hipLaunchKernelGGL(scanExclusive, dim3(1), dim3(1024), 0, s, keepPos, (uint32_t)m);
        hipLaunchKernelGGL(scanExclusive, dim3(1), dim3(1024), 0, s, mergePos, (uint32_t)m);
        RT_HIP(hipMemcpyAsync(&lastPos[0], keepPos + (m - 1), 4, hipMemcpyDeviceToHost, s)); RT_HIP(hipMemcpyAsync(&lastPos[1], mergePos + (m - 1), 4, hipMemcpyDeviceToHost, s));
        hipLaunchKernelGGL(plocScatter, g, dim3(256), 0, s, m, nodeBase, nn, keepPos, mergePos, clRef[a], clBox[a], clRef[a ^ 1], clBox[a ^ 1],
                           left, right, nodeParent, leafParent, nodeBox);
        RT_HIP(hipStreamSynchronize(s));
        const int kept = (int)(lastPos[0] + lastFlags[0]), merged = (int)(lastPos[1] + lastFlags[1]);
        if (merged <= 0 || kept != m - merged) { setError("buildBvh: clustering made no progress (%d clusters, %d merges, %d kept)", m, merged, kept); return -3; }
        roundBase.push_back((uint32_t)nodeBase);
        nodeBase += merged; m = kept; a ^= 1;
      }
      roundBase.push_back((uint32_t)nodeBase);
      c->mesh[slot].roundBase = roundBase;
      root = (int32_t)n - 2;                      // the last node created
      const int32_t none = -1;
      RT_HIP(hipMemcpyAsync(nodeParent + root, &none, 4, hipMemcpyHostToDevice, s));
      RT_HIP(hipStreamSynchronize(s));
      hipFree(clRef[0]); hipFree(clRef[1]); hipFree(nn); hipFree(clBox[0]); hipFree(clBox[1]); hipFree(keepPos); hipFree(mergePos);
    }
    hipLaunchKernelGGL(emitNodes, dim3(nb), dim3(256), 0, s, (int)n, order[cur], triBox, left, right, nodeBox, m.nodes);
    hipLaunchKernelGGL(emitNodes4, dim3(nb), dim3(256), 0, s, (int)n, order[cur], triBox, left, right, nodeParent, nodeBox, m.nodes4);
    hipLaunchKernelGGL(depthKernel, dim3(nb), dim3(256), 0, s, (int)n, nodeParent, leafParent, arrive + n);
    m.root = root;
  }
  RT_HIP(hipGetLastError());
  RT_HIP(hipMemcpyAsync(&m.depth, arrive + n, 4, hipMemcpyDeviceToHost, s));
  // the topology and the per-primitive boxes stay for rtggx_refit_as (freed with the mesh or by the next build)
  m.order = order[cur]; m.left = left; m.right = right; m.nodeParent = nodeParent; m.leafParent = leafParent; m.nodeBox = nodeBox; m.triBox = triBox;
  if (n > 1) {
    std::vector<int32_t> hl(n - 1), hr(n - 1);
    RT_HIP(hipMemcpyAsync(hl.data(), left, 4 * (size_t)(n - 1), hipMemcpyDeviceToHost, s)); RT_HIP(hipMemcpyAsync(hr.data(), right, 4 * (size_t)(n - 1), hipMemcpyDeviceToHost, s));
    RT_HIP(hipStreamSynchronize(s));
    { const int r = planTop(m, slot == 0 ? RT_TOP_SLOT0 : RT_TOP_SLOT1, hl, hr, s); if (r) return r; }
    if (m.roundBase.size() >= 2) {      // PLOC build: plan the refit (rtggx_refit_as) while the topology is at hand
      const int r = planRefit(m, hl, hr, s);
      if (r) return r;
    }
  }
  if (n > 1) { const int r = launchTreeCost(m, s); if (r) return r; RT_HIP(hipMemcpyAsync(m.hCost, m.dCost, 4, hipMemcpyDeviceToHost, s)); }
  RT_HIP(hipStreamSynchronize(s));
  if (n > 1) { m.builtCost = m.lastCost = *m.hCost; m.costInFlight = false; }
  hipFree(codes[0]); hipFree(codes[1]); hipFree(order[cur ^ 1]); hipFree(hist); hipFree(arrive);
  return 0;
}

}  // namespace rt
