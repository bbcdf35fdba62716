// Exhaustive / randomised check that the short division and square-root sequences used by temporalKernel
// (raytracedggx_amd/csrc/rtggx_device.h: divBy9, div3Shared, sqrtRN) return the IEEE round-to-nearest result,
// i.e. exactly what the compiler's own expansion of `/` and sqrtf returns.  Run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I raytracedggx_amd/csrc -I include tools/microbench/exactmath.hip -o /tmp/exactmath && /tmp/exactmath
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "rtggx_device.h"

__global__ void checkAll(unsigned long long* bad, uint32_t* firstBad) {
  const uint32_t stride = gridDim.x * blockDim.x;
  unsigned long long b9 = 0, bs = 0;
  for (uint64_t i = blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
    const float x = __uint_as_float((uint32_t)i);
    const float ax = fabsf(x);
    const bool normalRange = ax == 0.0f || (ax >= 1e-30f && ax <= 1e30f);
    if (!normalRange || (uint32_t)i == 0x80000000u) continue;     // divBy9(-0) = +0: the one difference, and no user cares about the sign
    const float q = rt::divBy9(x), qr = x / 9.0f;
    if (__float_as_uint(q) != __float_as_uint(qr)) { if (!b9) firstBad[0] = (uint32_t)i; ++b9; }
    if (x >= 0.0f) {
      const float s = rt::sqrtRN(x), sr = sqrtf(x);
      if (__float_as_uint(s) != __float_as_uint(sr)) { if (!bs) firstBad[1] = (uint32_t)i; ++bs; }
    }
  }
  if (b9) atomicAdd(&bad[0], b9);
  if (bs) atomicAdd(&bad[1], bs);
}

__device__ uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s; }

// tssTM's shape: numerators of any sign and size, denominator 4 + Y with Y >= 0 (and a sweep of other denominators)
__global__ void checkDiv3(unsigned long long* bad, uint32_t* firstBad, int rounds) {
  uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
  unsigned long long b = 0;
  for (int r = 0; r < rounds; ++r) {
    // exponent in [-40, 40], random mantissa and sign
    auto rnd = [&](int elo, int ehi) { const uint32_t m = lcg(s) >> 9; const int e = elo + (int)(lcg(s) >> 8) % (ehi - elo + 1); return __uint_as_float(((uint32_t)(e + 127) << 23) | m) ; };
    float d = (r & 1) ? 4.0f + rnd(-30, 20) : rnd(-20, 40);
    float n0 = rnd(-40, 30), n1 = -rnd(-40, 30), n2 = (r & 2) ? 0.0f : rnd(-40, 30);
    float q0, q1, q2;
    rt::div3Shared(n0, n1, n2, d, q0, q1, q2);
    if (__float_as_uint(q0) != __float_as_uint(n0 / d) || __float_as_uint(q1) != __float_as_uint(n1 / d) || __float_as_uint(q2) != __float_as_uint(n2 / d)) {
      if (!b) { firstBad[2] = __float_as_uint(n0); firstBad[3] = __float_as_uint(d); }
      ++b;
    }
  }
  if (b) atomicAdd(&bad[2], b);
}

int main() {
  unsigned long long* bad; uint32_t* first;
  hipMalloc(&bad, 32); hipMalloc(&first, 32); hipMemset(bad, 0, 32); hipMemset(first, 0, 32);
  hipLaunchKernelGGL(checkAll, dim3(4096), dim3(256), 0, 0, bad, first);
  hipLaunchKernelGGL(checkDiv3, dim3(4096), dim3(256), 0, 0, bad, first, 4096);
  unsigned long long h[4]; uint32_t f[8];
  hipMemcpy(h, bad, 32, hipMemcpyDeviceToHost); hipMemcpy(f, first, 32, hipMemcpyDeviceToHost);
  printf("x/9 over all floats with |x| in {0} u [1e-30, 1e30]: %llu mismatches (first %08x)\n", h[0], f[0]);
  printf("sqrt over all floats in {0} u [1e-30, 1e30]:          %llu mismatches (first %08x)\n", h[1], f[1]);
  printf("three quotients sharing a denominator, 4.3e9 random triples: %llu mismatches (first n %08x d %08x)\n", h[2], f[2], f[3]);
  return (h[0] || h[1] || h[2]) ? 1 : 0;
}
