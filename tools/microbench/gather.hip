// Microbenchmark: dependent random 64-byte record gathers (the access pattern of BVH traversal).
//   mode 0: every lane loads its own record with four 16-byte loads
//   mode 1: wave-cooperative: 4 lanes fetch one record (64 contiguous bytes), staged through LDS
// Each lane follows a pointer chain (next index stored in the record) for `steps` steps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <random>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) chase(const float4* __restrict__ table, const uint32_t* __restrict__ start, uint32_t* __restrict__ out, int steps) {
  __shared__ __attribute__((aligned(16))) char stageMem[4 * 64 * 80];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  char* stage = stageMem + wave * 64 * 80;
  uint32_t cur = start[blockIdx.x * 256 + threadIdx.x];
  float acc = 0.0f;
  for (int s = 0; s < steps; ++s) {
    float4 n0, n1, n2, n3;
    if (MODE == 0) {
      const float4* p = table + (size_t)cur * 4;
      n0 = p[0]; n1 = p[1]; n2 = p[2]; n3 = p[3];
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t src = (lane >> 2) + 16u * k;
        const uint32_t w = (uint32_t)__shfl((int)cur, (int)src);
        const float4 v = table[(size_t)w * 4 + (lane & 3u)];
        *reinterpret_cast<float4*>(stage + src * 80 + (lane & 3u) * 16) = v;
      }
      const float4* m = reinterpret_cast<const float4*>(stage + lane * 80);
      n0 = m[0]; n1 = m[1]; n2 = m[2]; n3 = m[3];
    }
    acc += n0.x + n1.y + n2.z;
    cur = __float_as_uint(n3.x);
  }
  out[blockIdx.x * 256 + threadIdx.x] = cur + (uint32_t)acc;
}

int main() {
  int cus = 256;
  const int grid = cus * 4, threads = grid * 256, steps = 64;
  for (size_t mb : {1, 2, 4, 8, 16, 64, 256}) {
    const size_t n = mb * 1024 * 1024 / 64;
    std::vector<uint32_t> perm(n);
    for (size_t i = 0; i < n; ++i) perm[i] = (uint32_t)i;
    std::mt19937 rng(1);
    std::shuffle(perm.begin(), perm.end(), rng);
    std::vector<float> host(n * 16, 0.5f);
    for (size_t i = 0; i < n; ++i) { uint32_t nx = perm[(i + 1) % n]; memcpy(&host[(size_t)perm[i] * 16 + 12], &nx, 4); }   // one big cycle
    std::vector<uint32_t> st(threads);
    for (int coherent = 0; coherent < 2; ++coherent) {
      for (int i = 0; i < threads; ++i) st[i] = coherent ? perm[((size_t)(i / 64) * 977) % n] : perm[((size_t)i * 7919) % n];   // coherent: all lanes of a wave walk the same chain
      float4* dT; uint32_t *dS, *dO;
      CHECK(hipMalloc(&dT, n * 64)); CHECK(hipMalloc(&dS, threads * 4)); CHECK(hipMalloc(&dO, threads * 4));
      CHECK(hipMemcpy(dT, host.data(), n * 64, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dS, st.data(), threads * 4, hipMemcpyHostToDevice));
      for (int mode = 0; mode < 2; ++mode) {
        hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
        for (int rep = 0; rep < 3; ++rep) {
          CHECK(hipEventRecord(a));
          if (mode == 0) hipLaunchKernelGGL(chase<0>, dim3(grid), dim3(256), 0, 0, dT, dS, dO, steps);
          else hipLaunchKernelGGL(chase<1>, dim3(grid), dim3(256), 0, 0, dT, dS, dO, steps);
          CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        }
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        const double lane_steps = (double)threads * steps;
        printf("table %4zu MB  %s  mode %d: %8.1f us  %6.2f G lane-steps/s  %7.1f GB/s useful  (per-wave step %.0f ns)\n", mb, coherent ? "wave-coherent" : "divergent    ", mode,
               ms * 1e3, lane_steps / ms / 1e6, lane_steps * 64 / ms / 1e6, ms * 1e6 / steps);
      }
      hipFree(dT); hipFree(dS); hipFree(dO);
    }
  }
  return 0;
}
