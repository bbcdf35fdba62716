// How close to the true x^512 / x^32 (the spatial filters' normal weights, SpatialFilter.hlsli:57-83) do the candidate fp32 formulations get?
// Every float x in [0.5, 1.25] is evaluated on the GPU and compared with pow((double)x, n) rounded to float; reported per band of the
// result's size (a weight below 1e-6 cannot move a filtered pixel), in ulps of the result.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/microbench/pow512.hip -o tools/microbench/pow512 && tools/microbench/pow512
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>

#define NCAND 7
// 0: nine / five squarings (the round-3 contract)   1: exp2(n * v_log_f32(x))   2: exp2(n * log2 via the series below)
// 3: squarings in two-float arithmetic              4: as 2, the product n * log2 carried as hi + lo into the exponential
__device__ __forceinline__ float log2Near1(float x) {      // log2(x) = log2e * (d - d^2/2 + ... ) with d = x - 1 exact for x in [0.5, 2]
  const float d = x - 1.0f;
  float p = -1.0f / 6.0f;
  p = __builtin_fmaf(p, d, 1.0f / 5.0f); p = __builtin_fmaf(p, d, -1.0f / 4.0f); p = __builtin_fmaf(p, d, 1.0f / 3.0f);
  p = __builtin_fmaf(p, d, -0.5f);
  const float q = __builtin_fmaf(p * d, d, d);      // d + d^2 p: the leading term is exact
  return q * 1.44269504088896341f;
}
__device__ float cand(int which, float x, int squarings) {
  const float n = (float)(1 << squarings);
  if (which == 0) { float p = x; for (int i = 0; i < squarings; ++i) p *= p; return p; }
  if (which == 1) return __builtin_amdgcn_exp2f(n * __builtin_amdgcn_logf(x));
  if (which == 2) return __builtin_amdgcn_exp2f(n * log2Near1(x));
  if (which == 3) {
    float h = x, l = 0.0f;
    for (int i = 0; i < squarings; ++i) { const float p = h * h; const float e = __builtin_fmaf(h, h, -p); l = __builtin_fmaf(2.0f * h, l, e); h = p; }
    return h + l;
  }
  if (which == 4) {
  const float d = x - 1.0f;
  float p = -1.0f / 6.0f;
  p = __builtin_fmaf(p, d, 1.0f / 5.0f); p = __builtin_fmaf(p, d, -1.0f / 4.0f); p = __builtin_fmaf(p, d, 1.0f / 3.0f); p = __builtin_fmaf(p, d, -0.5f);
  const float q = __builtin_fmaf(p * d, d, d);
  const float k = n * 1.44269504088896341f;
  const float hi = q * k, lo = __builtin_fmaf(q, k, -hi);
  return __builtin_amdgcn_exp2f(hi) * __builtin_fmaf(lo, 0.693147181f, 1.0f);
  }
  // 5: the hardware log2 of the ROUNDED 1 + d plus the rounding's correction (what an exact d = x - 1 allows); 6: only v_log_f32(x) * n, x exact
  const float d = x - 1.0f;      // exact here (x is the float); in the filter d comes from an exact integer dot product
  const float xr = 1.0f + d;
  const float delta = (1.0f - xr) + d;
  if (which == 5) return __builtin_amdgcn_exp2f(n * __builtin_fmaf(delta, 1.44269504088896341f, __builtin_amdgcn_logf(xr)));
  return __builtin_amdgcn_exp2f(n * __builtin_amdgcn_logf(x));
}
struct Acc { double maxUlp[3]; unsigned long long count[3]; };      // bands: result >= 1e-2, >= 1e-6, rest above 1e-30; in the second sweep (x <= 1.004 only): maxUlp[2] = the largest ABSOLUTE error x 2^24
__global__ void sweep(uint32_t lo, uint32_t hi, int squarings, Acc* out, int physical) {
  Acc a[NCAND] = {};
  for (uint32_t u = lo + blockIdx.x * blockDim.x + threadIdx.x; u < hi; u += gridDim.x * blockDim.x) {
    const float x = __uint_as_float(u);
    const double t = pow((double)x, (double)(1 << squarings));
    if (!(t > 1e-30) || t > 1e30) continue;
    const int band = t >= 1e-2 ? 0 : t >= 1e-6 ? 1 : 2;
    const float tf = (float)t;
    const double ulp = (double)(__uint_as_float(__float_as_uint(tf) + 1u) - tf);
    for (int c = 0; c < NCAND; ++c) {
      const double err = fabs((double)cand(c, x, squarings) - t);
      const double e = err / ulp;
      if (physical && band == 2) { if (err * 16777216.0 > a[c].maxUlp[2]) a[c].maxUlp[2] = err * 16777216.0; }
      else if (e > a[c].maxUlp[band]) a[c].maxUlp[band] = e;
      if (physical && err * 16777216.0 > a[c].maxUlp[2]) a[c].maxUlp[2] = err * 16777216.0;
      ++a[c].count[band];
    }
  }
  for (int c = 0; c < NCAND; ++c) for (int b = 0; b < 3; ++b) {
    atomicMax((unsigned long long*)&out[c].maxUlp[b], (unsigned long long)__double_as_longlong(a[c].maxUlp[b]));      // non-negative doubles order like their bits
    atomicAdd(&out[c].count[b], a[c].count[b]);
  }
}
// cost: a dependent chain would hide nothing; 32 independent evaluations per thread, all waves busy
__global__ void cost(int which, int squarings, float* out, int rounds) {
  float acc = 0.0f; float x = 0.97f + 1e-6f * (float)threadIdx.x;
  for (int r = 0; r < rounds; ++r) { acc += cand(which, x, squarings); x += 1e-7f; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
  Acc* d; hipMalloc(&d, sizeof(Acc) * NCAND);
  const char* names[NCAND] = {"squarings (r03 contract)", "exp2(n * v_log_f32 x)", "exp2(n * series log2)", "two-float squarings", "series log2, hi+lo exponent", "v_log_f32(1 + d) + rounding corr.", "v_log_f32(x) * n (= 1)"};
  for (int squarings : {9, 5}) {
    hipMemset(d, 0, sizeof(Acc) * NCAND);
    union { float f; uint32_t u; } lo, hi; lo.f = 0.5f; hi.f = 1.25f;
    hipLaunchKernelGGL(sweep, dim3(2048), dim3(256), 0, 0, lo.u, hi.u, squarings, d, 0);
    Acc h[NCAND]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("x^%d over every float in [0.5, 1.25], error in ulps of the correctly rounded result (max), by size of the result\n", 1 << squarings);
    for (int c = 0; c < NCAND; ++c) printf("  %-30s  >= 1e-2: %9.1f   1e-6 .. 1e-2: %9.1f   below: %11.1f\n", names[c], h[c].maxUlp[0], h[c].maxUlp[1], h[c].maxUlp[2]);
  }
  for (int squarings : {9, 5}) {      // the dot products unit normals can give: x <= 1.004
    hipMemset(d, 0, sizeof(Acc) * NCAND);
    union { float f; uint32_t u; } lo, hi; lo.f = 0.5f; hi.f = 1.004f;
    hipLaunchKernelGGL(sweep, dim3(2048), dim3(256), 0, 0, lo.u, hi.u, squarings, d, 1);
    Acc h[NCAND]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("x^%d over every float in [0.5, 1.004]: max error in ulps of the result by its size, and the largest ABSOLUTE error in units of 2^-24 (what a tap adds to a weight sum >= 1)\n", 1 << squarings);
    for (int c = 0; c < NCAND; ++c) printf("  %-34s  >= 1e-2: %9.1f   1e-6 .. 1e-2: %9.1f   absolute: %9.2f\n", names[c], h[c].maxUlp[0], h[c].maxUlp[1], h[c].maxUlp[2]);
  }
  float* o; hipMalloc(&o, 4 * 1024 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int squarings : {9, 5}) for (int c = 0; c < NCAND; ++c) {
    hipLaunchKernelGGL(cost, dim3(4096), dim3(256), 0, 0, c, squarings, o, 256);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(cost, dim3(4096), dim3(256), 0, 0, c, squarings, o, 4096);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("cost x^%d  %-30s %.3f ms for 4.3e9 evaluations\n", 1 << squarings, names[c], ms);
  }
  return 0;
}
