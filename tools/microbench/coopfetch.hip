// A traversal step's node fetch, two ways (round 3).  Every lane follows its own chain through 128-byte records (L2-resident, 8 MB), as a
// ray walks its own nodes; per step it needs the record's first seven 16-byte words.
//   own      each lane issues seven 16-byte loads from ITS record: 7 instructions x 64 distinct cache lines.  What traceKernel does; the
//            vector L1 takes about one line per cycle (lanecost.hip), so a wave-step occupies it for ~450 cycles.
//   coop     the 8 lanes of a group fetch the records of the group's 8 rays one after the other: in instruction k lane 8g + p loads word p
//            of the record of ray 8g + k -- 8 instructions x 8 distinct lines, each line read whole by 8 lanes.  The words reach their
//            rays through LDS: lane 8g + p writes word p of ray 8g + k to tile[p][8g + k], ray r reads tile[0..6][r].
//            64 lines per step instead of 448, for 8 LDS writes + 7 LDS reads of 16 bytes per lane and 8 cross-lane broadcasts of the
//            record index (ds_bpermute).
// WPC workgroups of 256 threads per CU (4 -> 16 waves per CU, 1 -> 4 waves per CU: the trace kernel keeps 12 per CU resident).
// Build: hipcc -O3 --offload-arch=gfx950 coopfetch.hip -o coopfetch
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) chaseOwn(const uint32_t* __restrict__ table, const uint32_t* __restrict__ start, uint32_t* __restrict__ out, int steps, unsigned long long laneMask) {
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t cur = start[blockIdx.x * 256 + threadIdx.x], acc = 0;
  if ((laneMask >> lane) & 1ull) {
    for (int s = 0; s < steps; ++s) {
      const uint4* p = reinterpret_cast<const uint4*>(table + (size_t)cur * 32);
      uint4 v[7];
#pragma unroll
      for (int k = 0; k < 7; ++k) v[k] = p[k];
#pragma unroll
      for (int k = 1; k < 7; ++k) acc += v[k].x + v[k].y + v[k].z + v[k].w;
      acc += v[0].y + v[0].z + v[0].w;
      cur = v[0].x;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = cur + acc;
}

// tile[p][r]: word p of ray r, 16 bytes each; lanes of a read (fixed p, r = lane) are consecutive: conflict-free.  A write has fixed k and
// lanes (g, p) -> tile[p][8 g + k]: the 8 lanes of a group hit 8 different rows p, the groups are 8 entries apart -- rows are padded so
// that the eight rows of a write start in different banks.
#define TILE_STRIDE (64 + 1)
__global__ void __launch_bounds__(256) chaseCoop(const uint32_t* __restrict__ table, const uint32_t* __restrict__ start, uint32_t* __restrict__ out, int steps, unsigned long long laneMask) {
  __shared__ uint4 tileMem[4][8 * TILE_STRIDE];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, p = lane & 7u, g8 = lane & ~7u;
  uint4* const tile = tileMem[wave];
  uint32_t cur = start[blockIdx.x * 256 + threadIdx.x], acc = 0;
  const bool active = (laneMask >> lane) & 1ull;
  for (int s = 0; s < steps; ++s) {
    const unsigned long long need = __ballot(active);
    uint4 w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t ck = (uint32_t)__shfl((int)cur, (int)(g8 | (uint32_t)k));      // the record of ray 8g + k
      const bool on = (need >> (g8 | (uint32_t)k)) & 1ull;                          // uniform per group of 8 lanes
      w[k] = on ? reinterpret_cast<const uint4*>(table + (size_t)ck * 32)[p] : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) tile[p * TILE_STRIDE + (g8 | (uint32_t)k)] = w[k];
    __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): the wave's own LDS writes have landed (one wave: no barrier needed)
    __builtin_amdgcn_wave_barrier();
    if (active) {
      uint4 v[7];
#pragma unroll
      for (int k = 0; k < 7; ++k) v[k] = tile[k * TILE_STRIDE + lane];
#pragma unroll
      for (int k = 1; k < 7; ++k) acc += v[k].x + v[k].y + v[k].z + v[k].w;
      acc += v[0].y + v[0].z + v[0].w;
      cur = v[0].x;
    }
    __builtin_amdgcn_wave_barrier();
  }
  out[blockIdx.x * 256 + threadIdx.x] = cur + acc;
}

template <typename K> static float run(K kernel, const uint32_t* dT, const uint32_t* dS, uint32_t* dO, int grid, int steps, unsigned long long mask) {
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, dT, dS, dO, steps, mask);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    CHECK(hipEventElapsedTime(&ms, a, b));
  }
  return ms;
}

int main() {
  const int cus = 256, steps = 64;
  const size_t n = 8u * 1024 * 1024 / 128;
  std::vector<uint32_t> perm(n);
  for (size_t i = 0; i < n; ++i) perm[i] = (uint32_t)i;
  std::mt19937 rng(1);
  std::shuffle(perm.begin(), perm.end(), rng);
  std::vector<uint32_t> host(n * 32);
  for (size_t i = 0; i < n * 32; ++i) host[i] = (uint32_t)(i * 2654435761u);
  for (size_t i = 0; i < n; ++i) host[(size_t)perm[i] * 32] = perm[(i + 1) % n];   // one big cycle, next index in word 0
  const int maxThreads = cus * 4 * 256;
  std::vector<uint32_t> st(maxThreads);
  for (int i = 0; i < maxThreads; ++i) st[i] = perm[((size_t)i * 7919) % n];
  uint32_t *dT, *dS, *dO;
  CHECK(hipMalloc(&dT, n * 128)); CHECK(hipMalloc(&dS, maxThreads * 4)); CHECK(hipMalloc(&dO, maxThreads * 4));
  CHECK(hipMemcpy(dT, host.data(), n * 128, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dS, st.data(), maxThreads * 4, hipMemcpyHostToDevice));
  // same results?
  {
    std::vector<uint32_t> a(maxThreads), b(maxThreads);
    run(chaseOwn, dT, dS, dO, cus * 4, steps, ~0ull); CHECK(hipMemcpy(a.data(), dO, maxThreads * 4, hipMemcpyDeviceToHost));
    run(chaseCoop, dT, dS, dO, cus * 4, steps, ~0ull); CHECK(hipMemcpy(b.data(), dO, maxThreads * 4, hipMemcpyDeviceToHost));
    printf("results %s\n", a == b ? "identical" : "DIFFER");
  }
  struct M { const char* name; unsigned long long mask; } masks[] = {
    {"64 lanes      ", ~0ull}, {"first 32      ", 0xFFFFFFFFull}, {"every 2nd (32)", 0x5555555555555555ull}, {"first 16      ", 0xFFFFull}, {"every 4th (16)", 0x1111111111111111ull}, {"every 8th (8) ", 0x0101010101010101ull}};
  for (int wpc : {4, 3, 1}) {
    printf("%d waves per CU, %d steps; ns per wave-step (kernel time / steps)\n", wpc * 4, steps);
    for (const M& m : masks) {
      const float a = run(chaseOwn, dT, dS, dO, cus * wpc, steps, m.mask), b = run(chaseCoop, dT, dS, dO, cus * wpc, steps, m.mask);
      printf("  %s  own %6.0f   coop %6.0f\n", m.name, a * 1e6 / steps, b * 1e6 / steps);
    }
  }
  return 0;
}
