// Microbenchmark: what does one wave-level instruction cost a gfx950 SIMD, as a function of how many waves share the SIMD?
//
// The question behind it (DESIGN.md "Roofline accounting"): the frame issues ~108 M wave-level VALU instructions;
// priced at 4 cycles each that is 81 % of the issue slots of 1024 SIMDs over a 216 us frame ("VALU-issue-bound"), priced at 2
// (MI355X_MICROARCH.md constants table: v_fma_f32 wave64 2 cycles on a SIMD-32, 4 for a wave that is alone) it is 41 %.
//
// Method.  Every CU gets the same number of resident waves, W per SIMD (one or two workgroups per CU; 80 KB of LDS per
// workgroup caps a CU at two, the HW_ID of every wave is recorded and the actual waves-per-SIMD histogram is printed).  Each wave
// runs ITER iterations of a block of 256 instructions of one kind, written as volatile inline asm so that the compiler neither
// reorders nor removes them, and stamps s_memtime before and after.  Reported per stream and per W:
//   per-wave  = median (t1 - t0) / instructions        cycles one wave needs per instruction (latency view)
//   per-SIMD  = per-wave / waves on that SIMD           cycles of SIMD time per wave-level instruction (throughput view)
// Streams: independent v_fma_f32 (8 accumulators), dependent v_fma_f32 (1 accumulator), v_exp_f32, v_rcp_f32, v_mul_lo_u32,
// v_fma_f64, v_cndmask/v_cmp pairs, ds_bpermute_b32 (the __shfl of the traversal's work sharing), s_add_u32 (SALU, 8 chains),
// and the traversal kernel's mix (5 VALU : 2 SALU, trace.hip: ~250 VALU + ~100 SALU per wave-step).
//
// Build: hipcc -O2 --offload-arch=gfx950 tools/microbench/valu_issue.hip -o tools/microbench/valu_issue
// Run  : tools/microbench/valu_issue > profiles/r02_a_valu_issue.txt
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

enum Stream { FMA_INDEP, FMA_DEP, EXP, RCP, MUL_LO, FMA64, CNDMASK, BPERMUTE, SALU, MIX, NUM_STREAMS };
static const char* kNames[NUM_STREAMS] = {"v_fma_f32 x8 independent", "v_fma_f32 dependent chain", "v_exp_f32", "v_rcp_f32", "v_mul_lo_u32",
                                          "v_fma_f64", "v_cmp_lt_f32 + v_cndmask_b32", "ds_bpermute_b32 (+wait per 8)", "s_add_u32 x8 independent",
                                          "mix 5 VALU : 2 SALU (traversal step)"};

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

struct Stamp { unsigned long long t0, t1; uint32_t hwId, xcc; };

template <int S>
__global__ void __launch_bounds__(1024) issueKernel(Stamp* __restrict__ out, float* __restrict__ sink, int iters, float seed) {
  extern __shared__ uint32_t lds[];     // only to cap the workgroups per CU
  float a0 = seed, a1 = seed + 1.0f, a2 = seed + 2.0f, a3 = seed + 3.0f, a4 = seed + 4.0f, a5 = seed + 5.0f, a6 = seed + 6.0f, a7 = seed + 7.0f;
  const float x = 1.0000001f, y = 1e-9f;
  double d0 = seed, d1 = seed + 1.0, d2 = seed + 2.0, d3 = seed + 3.0;
  const double dx = 1.0000001, dy = 1e-9;
  uint32_t u0 = threadIdx.x, u1 = threadIdx.x + 1, u2 = threadIdx.x + 2, u3 = threadIdx.x + 3;
  uint32_t s0 = 1, s1 = 2, s2 = 3, s3 = 4, s4 = 5, s5 = 6, s6 = 7, s7 = 8;
  const uint32_t addr = ((threadIdx.x + 1) & 63u) * 4u;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (S == FMA_INDEP) {      // 32 x 8 = 256
      REP32(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));)
    } else if (S == FMA_DEP) {
      REP32(REP8(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(x), "v"(y));))
    } else if (S == EXP) {
      REP32(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (S == RCP) {
      REP32(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (S == MUL_LO) {
      REP32(REP4(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(addr | 1u));))
    } else if (S == FMA64) {
      REP32(REP4(asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(d0), "+v"(d1) : "v"(dx), "v"(dy));))
    } else if (S == CNDMASK) {
      REP32(REP4(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a0), "+v"(a1) : : "vcc");))
    } else if (S == BPERMUTE) {   // 8 shuffles in flight, then the wait the consumer needs: the pattern of the steal phase
      REP32(asm volatile("ds_bpermute_b32 %0, %8, %0\n ds_bpermute_b32 %1, %8, %1\n ds_bpermute_b32 %2, %8, %2\n ds_bpermute_b32 %3, %8, %3\n"
                         "ds_bpermute_b32 %4, %8, %4\n ds_bpermute_b32 %5, %8, %5\n ds_bpermute_b32 %6, %8, %6\n ds_bpermute_b32 %7, %8, %7\n s_waitcnt lgkmcnt(0)"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(addr));)
    } else if (S == SALU) {
      REP32(asm volatile("s_add_u32 %0, %0, %8\n s_add_u32 %1, %1, %8\n s_add_u32 %2, %2, %8\n s_add_u32 %3, %3, %8\n"
                         "s_add_u32 %4, %4, %8\n s_add_u32 %5, %5, %8\n s_add_u32 %6, %6, %8\n s_add_u32 %7, %7, %8"
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : "s"(3u) : "scc");)
    } else if (S == MIX) {     // 7 instructions per repeat: 5 VALU + 2 SALU; 36 repeats = 252 (counted exactly below)
      REP32(asm volatile("v_fma_f32 %0, %0, %6, %7\n v_fma_f32 %1, %1, %6, %7\n s_add_u32 %4, %4, 3\n v_fma_f32 %2, %2, %6, %7\n v_fma_f32 %3, %3, %6, %7\n s_add_u32 %5, %5, 5\n v_fma_f32 %0, %0, %6, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1) : "v"(x), "v"(y) : "scc");)
      REP4(asm volatile("v_fma_f32 %0, %0, %6, %7\n v_fma_f32 %1, %1, %6, %7\n s_add_u32 %4, %4, 3\n v_fma_f32 %2, %2, %6, %7\n v_fma_f32 %3, %3, %6, %7\n s_add_u32 %5, %5, 5\n v_fma_f32 %0, %0, %6, %7"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1) : "v"(x), "v"(y) : "scc");)
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63u) == 0u) {
    Stamp st; st.t0 = t0; st.t1 = t1;
    st.hwId = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));      // HW_REG_HW_ID, all 32 bits
    st.xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 15u;  // HW_REG_XCC_ID
    out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = st;
  }
  // keep every chain alive
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3) + (float)(u0 + u1 + u2 + u3 + s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7) == 12345.678f) sink[threadIdx.x] = a0;
  if (seed == 12345.0f) sink[1] = (float)lds[threadIdx.x];    // touches the LDS allocation so that it is kept
}

static int instructionsPerIteration(int s) { return s == MIX ? 36 * 7 : 256; }

template <int S>
static void run(int wavesPerSimd, int numCUs, Stamp* dOut, float* dSink, FILE* f) {
  // W <= 4: one workgroup of 4W waves per CU; W = 6, 8: two workgroups of 2W waves each.  80 KB of LDS per workgroup: at most two per CU.
  const int wgPerCU = wavesPerSimd > 4 ? 2 : 1;
  const int wavesPerWG = 4 * wavesPerSimd / wgPerCU;
  const int grid = numCUs * wgPerCU, block = wavesPerWG * 64;
  const int iters = 200;
  const size_t ldsBytes = 80 * 1024;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(issueKernel<S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(issueKernel<S>, dim3(grid), dim3(block), ldsBytes, 0, dOut, dSink, iters, 1.0f);
  CHECK(hipDeviceSynchronize());
  std::vector<Stamp> st((size_t)grid * wavesPerWG);
  CHECK(hipMemcpy(st.data(), dOut, st.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
  // waves per (xcc, se, sh, cu, simd): HW_ID bits simd 5:4, cu 11:8, sh 12, se 15:13 (gfx9 layout)
  std::map<uint32_t, int> perSimd;
  for (const Stamp& s : st) perSimd[(s.xcc << 16) | (s.hwId & 0xFF30u)]++;
  std::map<int, int> histo;
  for (auto& kv : perSimd) histo[kv.second]++;
  std::vector<double> perWave;
  const double insts = (double)iters * instructionsPerIteration(S);
  for (const Stamp& s : st) perWave.push_back((double)(s.t1 - s.t0) / insts);
  std::sort(perWave.begin(), perWave.end());
  const double med = perWave[perWave.size() / 2];
  fprintf(f, "%-40s W=%d  per-wave %7.2f cyc/instr (min %6.2f max %6.2f)  per-SIMD %6.2f cyc/wave-instr   SIMDs in use %zu, waves/SIMD histogram:", kNames[S], wavesPerSimd, med,
          perWave.front(), perWave.back(), med / wavesPerSimd, perSimd.size());
  for (auto& kv : histo) fprintf(f, " %dx%d", kv.second, kv.first);
  fprintf(f, "\n");
}

template <int S>
static void sweep(int numCUs, Stamp* dOut, float* dSink, FILE* f) {
  for (int w : {1, 2, 3, 4, 6, 8}) run<S>(w, numCUs, dOut, dSink, f);
  fprintf(f, "\n");
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int numCUs = prop.multiProcessorCount;
  Stamp* dOut; float* dSink;
  CHECK(hipMalloc(&dOut, sizeof(Stamp) * (size_t)numCUs * 64));
  CHECK(hipMalloc(&dSink, 4096));
  FILE* f = stdout;
  fprintf(f, "# %s, %d CUs, clock %d kHz; s_memtime ticks = shader cycles (MI355X_MICROARCH.md constants table)\n", prop.name, numCUs, prop.clockRate);
  fprintf(f, "# per-wave: cycles ONE wave needs per instruction; per-SIMD: per-wave / W = SIMD time per wave-level instruction when W waves share the SIMD\n\n");
  sweep<FMA_INDEP>(numCUs, dOut, dSink, f);
  sweep<FMA_DEP>(numCUs, dOut, dSink, f);
  sweep<EXP>(numCUs, dOut, dSink, f);
  sweep<RCP>(numCUs, dOut, dSink, f);
  sweep<MUL_LO>(numCUs, dOut, dSink, f);
  sweep<FMA64>(numCUs, dOut, dSink, f);
  sweep<CNDMASK>(numCUs, dOut, dSink, f);
  sweep<BPERMUTE>(numCUs, dOut, dSink, f);
  sweep<SALU>(numCUs, dOut, dSink, f);
  sweep<MIX>(numCUs, dOut, dSink, f);
  return 0;
}
