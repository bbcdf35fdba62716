// Microbenchmark: what does a divergent gather cost the vector L1 -- per instruction, per active lane, or per byte?
// Every active lane follows its own pointer chain through a table of 128-byte records (L2-resident, 8 MB); per step
// it issues NLOADS loads of WIDTH dwords from its record.  Lanes are masked off by lane index (ACTIVE of 64, either
// the first ACTIVE lanes or every (64/ACTIVE)-th lane).  16 waves per CU, all CUs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <random>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NLOADS, int WIDTH>
__global__ void __launch_bounds__(256) chase(const uint32_t* __restrict__ table, const uint32_t* __restrict__ start, uint32_t* __restrict__ out, int steps,
                                             unsigned long long laneMask, int share = 1) {
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t cur = start[blockIdx.x * 256 + threadIdx.x];
  // share = G: groups of G consecutive lanes follow the SAME chain (same record every step), as neighbouring rays walking the
  // same nodes do.  Does the vector L1 serve such a load as one request per distinct line, or one per lane?
  if (share > 1) cur = (uint32_t)__shfl((int)cur, (int)(lane & ~(uint32_t)(share - 1)));
  uint32_t acc = 0;
  if ((laneMask >> lane) & 1ull) {
    for (int s = 0; s < steps; ++s) {
      const uint32_t* p = table + (size_t)cur * 32;
      uint32_t nxt = 0;
#pragma unroll
      for (int k = 0; k < NLOADS; ++k) {
        if (WIDTH == 4) { const uint4 v = *reinterpret_cast<const uint4*>(p + 4 * k); acc += v.y + v.z + v.w; if (k == 0) nxt = v.x; else acc += v.x; }
        if (WIDTH == 2) { const uint2 v = *reinterpret_cast<const uint2*>(p + 2 * k); acc += v.y; if (k == 0) nxt = v.x; else acc += v.x; }
        if (WIDTH == 1) { const uint32_t v = p[k]; if (k == 0) nxt = v; else acc += v; }
      }
      cur = nxt;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = cur + acc;
}

template <int NLOADS, int WIDTH>
static float run(const uint32_t* dT, const uint32_t* dS, uint32_t* dO, int grid, int steps, unsigned long long mask, int share = 1) {
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((chase<NLOADS, WIDTH>), dim3(grid), dim3(256), 0, 0, dT, dS, dO, steps, mask, share);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    CHECK(hipEventElapsedTime(&ms, a, b));
  }
  return ms;
}

int main() {
  const int cus = 256, grid = cus * 4, threads = grid * 256, steps = 64;
  const size_t n = 8u * 1024 * 1024 / 128;
  std::vector<uint32_t> perm(n);
  for (size_t i = 0; i < n; ++i) perm[i] = (uint32_t)i;
  std::mt19937 rng(1);
  std::shuffle(perm.begin(), perm.end(), rng);
  std::vector<uint32_t> host(n * 32, 1u);
  for (size_t i = 0; i < n; ++i) host[(size_t)perm[i] * 32] = perm[(i + 1) % n];   // one big cycle, next index in word 0
  std::vector<uint32_t> st(threads);
  for (int i = 0; i < threads; ++i) st[i] = perm[((size_t)i * 7919) % n];
  uint32_t *dT, *dS, *dO;
  CHECK(hipMalloc(&dT, n * 128)); CHECK(hipMalloc(&dS, threads * 4)); CHECK(hipMalloc(&dO, threads * 4));
  CHECK(hipMemcpy(dT, host.data(), n * 128, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dS, st.data(), threads * 4, hipMemcpyHostToDevice));
  struct M { const char* name; unsigned long long mask; int active; } masks[] = {
    {"64 lanes      ", ~0ull, 64}, {"first 32      ", 0xFFFFFFFFull, 32}, {"every 2nd (32)", 0x5555555555555555ull, 32},
    {"first 16      ", 0xFFFFull, 16}, {"every 4th (16)", 0x1111111111111111ull, 16}, {"first 4       ", 0xFull, 4}, {"every 16th (4)", 0x0001000100010001ull, 4}};
  printf("16 waves/CU, %d steps; time per wave-step = kernel time / steps (all waves run concurrently)\n", steps);
  for (const M& m : masks) {
    const float a = run<1, 4>(dT, dS, dO, grid, steps, m.mask), b = run<4, 4>(dT, dS, dO, grid, steps, m.mask), c = run<7, 4>(dT, dS, dO, grid, steps, m.mask);
    const float d = run<1, 1>(dT, dS, dO, grid, steps, m.mask), e = run<4, 1>(dT, dS, dO, grid, steps, m.mask), f = run<8, 2>(dT, dS, dO, grid, steps, m.mask);
    printf("%s  ns per wave-step:  1 x16B %6.0f   4 x16B %6.0f   7 x16B %6.0f   1 x4B %6.0f   4 x4B %6.0f   8 x8B %6.0f\n", m.name,
           a * 1e6 / steps, b * 1e6 / steps, c * 1e6 / steps, d * 1e6 / steps, e * 1e6 / steps, f * 1e6 / steps);
  }
  printf("\nlanes sharing a record (all 64 lanes active; groups of G consecutive lanes read the same 128-byte record):\n");
  for (int g : {1, 2, 4, 8, 16, 64}) {
    const float a = run<1, 4>(dT, dS, dO, grid, steps, ~0ull, g), b = run<4, 4>(dT, dS, dO, grid, steps, ~0ull, g), c = run<7, 4>(dT, dS, dO, grid, steps, ~0ull, g);
    printf("G = %2d (%2d distinct records per load)  ns per wave-step:  1 x16B %6.0f   4 x16B %6.0f   7 x16B %6.0f\n", g, 64 / g, a * 1e6 / steps, b * 1e6 / steps, c * 1e6 / steps);
  }
  return 0;
}