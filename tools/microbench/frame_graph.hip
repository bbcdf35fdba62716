// Would ONE graph per frame -- a linear chain of the frame's kernels with external event nodes where a stage must follow the same stage
// of the frame before -- keep the three-stage pipeline of capi.hip going, and what would it cost the host?
//
// The frame of a 1920 x 171 strip (durations in us, profiles/r03_c_strip_chain.txt):
//     stage C   rasterSmall 13, rasterLarge 9, rayGen 10      serial across frames (ray generation reads what the one before wrote)
//     stage B   trace 47                                       alternates between two "streams" (two traversals in flight)
//     stage M   shade 11, H 8, V 10, temporal 10, tone map 5   serial across frames (history)
// Today: three streams + events, ~10 launches and ~3 waits per frame = 48 us of host time per frame, 60 us per frame on the GPU.
// Here: graph f = [wait C] C-kernels [record C] [wait B|R] trace [record B|R] [wait M] M-kernels [record M] [record frame end], launched
// on stream L[f % 4]; the events are ordinary hipEvents in external record / wait nodes (one per stage: a wait node sees the record the
// previous graph launch enqueued).  Variant "shade with the traversal": stage B = trace + shade, stage M = H, V, temporal, tone map.
// The stand-in kernels wait on the 100 MHz counter with 64 workgroups, so that any number of them overlap freely: the result is what the
// submission scheme allows, not what the real kernels' contention adds.
// Build: hipcc -O2 --offload-arch=gfx950 frame_graph.hip -o frame_graph
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Params { float v[228]; };      // 912 bytes by value, like FrameParams
__global__ void spin(float* p, int ticks, int tag) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = (float)tag;
}
__global__ void spinBig(float* p, int ticks, Params q) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = q.v[0];
}
// order check: every stage kernel of frame f must see the same stage of frame f - 1 finished (a counter per stage)
__global__ void stageCheck(unsigned int* counters, int stage, unsigned int frame, unsigned int* errors, int ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { if (counters[stage] != frame) atomicAdd(errors, 1u); }
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0 && blockIdx.x == 0) { __threadfence(); counters[stage] = frame + 1u; }
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const int frames = argc > 1 ? atoi(argv[1]) : 3000;
  float* d; CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
  unsigned int* cnt; CK(hipMalloc(&cnt, 64)); CK(hipMemset(cnt, 0, 64));
  Params P; for (int i = 0; i < 228; ++i) P.v[i] = 1.0f;
  const dim3 g(64), b(64);
  hipStream_t L[4]; for (auto& s : L) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t evC, evB[2], evM, evEnd[4];
  CK(hipEventCreateWithFlags(&evC, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&evM, hipEventDisableTiming));
  for (auto& e : evB) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (auto& e : evEnd) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));

  auto run = [&](const char* name, auto&& body) {
    for (int i = 0; i < 200; ++i) body(i);
    CK(hipDeviceSynchronize());
    double host = 0.0; const double t0 = now();
    for (int i = 0; i < frames; ++i) { const double a = now(); body(200 + i); host += now() - a; }
    CK(hipDeviceSynchronize());
    const double wall = now() - t0;
    printf("%-58s host %6.2f us per frame, wall %6.2f us per frame\n", name, host / frames * 1e6, wall / frames * 1e6);
  };

  // ---- graphs: 8 of them (launch stream f % 4 x traversal stage f % 2), kernels' arguments replaced per launch where they change
  for (int variant = 0; variant < 2; ++variant) {
    struct G { hipGraphExec_t exec; hipGraphNode_t first; std::vector<hipGraphNode_t> checks; };
    std::vector<G> graphs(8);
    for (int k = 0; k < 8; ++k) {
      hipGraph_t fg; CK(hipGraphCreate(&fg, 0));
      hipGraphNode_t prev = nullptr;
      auto dep = [&]() { return prev ? &prev : nullptr; };
      auto kernel = [&](int us, bool big, int stage) {
        hipGraphNode_t nd; hipKernelNodeParams kp = {}; kp.gridDim = g; kp.blockDim = b;
        float* pp = d + k * 16; int ticks = us * 100, tag = us; unsigned int fr = 0; unsigned int* errs = cnt + 8;
        void* a3[3] = {&pp, &ticks, &tag}; void* aB[3] = {&pp, &ticks, &P}; void* aS[5] = {&cnt, &stage, &fr, &errs, &ticks};
        if (stage >= 0) { kp.func = (void*)stageCheck; kp.kernelParams = aS; } else if (big) { kp.func = (void*)spinBig; kp.kernelParams = aB; } else { kp.func = (void*)spin; kp.kernelParams = a3; }
        CK(hipGraphAddKernelNode(&nd, fg, dep(), prev ? 1 : 0, &kp)); prev = nd;
        if (stage >= 0) graphs[k].checks.push_back(nd);
        return nd;
      };
      auto wait = [&](hipEvent_t e) { hipGraphNode_t nd; CK(hipGraphAddEventWaitNode(&nd, fg, dep(), prev ? 1 : 0, e)); prev = nd; };
      auto record = [&](hipEvent_t e) { hipGraphNode_t nd; CK(hipGraphAddEventRecordNode(&nd, fg, dep(), prev ? 1 : 0, e)); prev = nd; };
      hipEvent_t eB = evB[k & 1];
      wait(evC); graphs[k].first = kernel(13, true, -1); kernel(9, false, -1); kernel(10, false, 0); record(evC);
      wait(eB); kernel(47, false, 1 + (k & 1)); if (variant == 1) kernel(11, false, -1); record(eB);
      wait(evM); if (variant == 0) kernel(11, false, -1); kernel(8, false, -1); kernel(10, false, -1); kernel(10, false, 3); kernel(5, false, -1); record(evM); record(evEnd[k & 3]);
      CK(hipGraphInstantiate(&graphs[k].exec, fg, nullptr, nullptr, 0));
    }
    CK(hipMemset(cnt, 0, 64));
    unsigned int frameNo = 0;
    char name[96]; snprintf(name, sizeof name, "frame graphs, 4 launch streams%s", variant ? ", shade with the traversal" : "");
    run(name, [&](int) {
      const int k = (int)(frameNo & 7u);
      G& gr = graphs[k];
      if (frameNo >= 4) CK(hipEventSynchronize(evEnd[frameNo & 3]));      // the frames-in-flight fence
      // per-frame arguments: the 912-byte constants of the first kernel, the frame number of the order checks
      { float* pp = d + k * 16; int ticks = 1300; P.v[0] += 1.0f; void* aB[3] = {&pp, &ticks, &P};
        hipKernelNodeParams kp = {}; kp.func = (void*)spinBig; kp.gridDim = g; kp.blockDim = b; kp.kernelParams = aB; CK(hipGraphExecKernelNodeSetParams(gr.exec, gr.first, &kp)); }
      const int stages[3] = {0, 1 + (k & 1), 3}; const int us[3] = {10, 47, 10};
      for (int j = 0; j < 3; ++j) {
        unsigned int fr = stages[j] == 1 || stages[j] == 2 ? frameNo / 2 : frameNo; unsigned int* errs = cnt + 8; int st = stages[j], ticks = us[j] * 100;
        void* aS[5] = {&cnt, &st, &fr, &errs, &ticks};
        hipKernelNodeParams kp = {}; kp.func = (void*)stageCheck; kp.gridDim = g; kp.blockDim = b; kp.kernelParams = aS; CK(hipGraphExecKernelNodeSetParams(gr.exec, gr.checks[j], &kp));
      }
      CK(hipGraphLaunch(gr.exec, L[frameNo & 3])); ++frameNo;
    });
    unsigned int errors = 0; CK(hipMemcpy(&errors, cnt + 8, 4, hipMemcpyDeviceToHost));
    printf("    stage-order violations seen by the kernels: %u of %u checks\n", errors, 3 * frameNo);
  }

  // ---- today's submission: three streams (+ the second traversal stream), events riding on kernels
  {
    hipStream_t sC = L[0], sB[2] = {L[1], L[2]}, sM = L[3];
    hipEvent_t eGen, eTr[2], eSet[4];
    CK(hipEventCreateWithFlags(&eGen, hipEventDisableTiming)); for (auto& e : eTr) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); for (auto& e : eSet) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    unsigned int frameNo = 0;
    for (int variant = 0; variant < 2; ++variant) {
      char name[96]; snprintf(name, sizeof name, "streams + events (capi.hip today)%s", variant ? ", shade with the traversal" : "");
      frameNo = 0;
      run(name, [&](int) {
        const int k = (int)(frameNo & 1u);
        if (frameNo >= 4) CK(hipEventSynchronize(eSet[frameNo & 3]));
        P.v[0] += 1.0f;
        hipLaunchKernelGGL(spinBig, g, b, 0, sC, d, 1300, P);
        hipLaunchKernelGGL(spin, g, b, 0, sC, d, 900, 1);
        hipExtLaunchKernelGGL(spin, g, b, 0, sC, nullptr, eGen, 0, d, 1000, 2);
        CK(hipStreamWaitEvent(sB[k], eGen, 0));
        if (variant == 0) hipExtLaunchKernelGGL(spin, g, b, 0, sB[k], nullptr, eTr[k], 0, d + 16, 4700, 3);
        else { hipLaunchKernelGGL(spin, g, b, 0, sB[k], d + 16, 4700, 3); hipExtLaunchKernelGGL(spin, g, b, 0, sB[k], nullptr, eTr[k], 0, d + 16, 1100, 4); }
        CK(hipStreamWaitEvent(sM, eTr[k], 0));
        if (variant == 0) hipLaunchKernelGGL(spin, g, b, 0, sM, d + 32, 1100, 4);
        hipLaunchKernelGGL(spin, g, b, 0, sM, d + 32, 800, 5);
        hipLaunchKernelGGL(spin, g, b, 0, sM, d + 32, 1000, 6);
        hipLaunchKernelGGL(spin, g, b, 0, sM, d + 32, 1000, 7);
        hipExtLaunchKernelGGL(spin, g, b, 0, sM, nullptr, eSet[frameNo & 3], 0, d + 32, 500, 8);
        ++frameNo;
      });
    }
  }
  return 0;
}
