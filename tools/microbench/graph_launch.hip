// What does a frame's submission cost the HOST on this ROCm, launched as individual kernels against replayed as a captured graph?
// The frame of librtggx on a thin strip is ~10 short kernels over three streams joined by events (capi.hip); the host pays ~4 us per
// launch (tools/probes/host_cost_probe.py).  This microbenchmark issues a stand-in with the same shape -- chain C (4 kernels) -> chain B
// (1 kernel) -> chain M (5 kernels), tiny kernels -- in four ways and prints the host time per "frame" and the wall time per frame:
//   streams     hipLaunchKernelGGL / hipExtLaunchKernelGGL + hipStreamWaitEvent, as capi.hip does today
//   graph       the same sequence captured once (hipStreamBeginCapture on C, forked to B and M by events), replayed with hipGraphLaunch
//   graph+set   as above, with the kernel arguments of K nodes replaced before every launch (hipGraphExecKernelNodeSetParams)
//   one stream  all ten kernels on one stream, no events (lower bound of the launch path)
// Build: hipcc -O2 --offload-arch=gfx950 graph_launch.hip -o graph_launch
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Params { float v[200]; };      // ~800 bytes by value, like FrameParams
__global__ void tiny(float* p, int n, int tag) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + (float)tag; }
__global__ void tinyBig(float* p, int n, Params q) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += q.v[i % 200]; }

// a stand-in of given duration that leaves the chip free for others: 64 workgroups wait on the 100 MHz counter
__global__ void spin(float* p, int ticks, int tag) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = (float)tag;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const int frames = argc > 1 ? atoi(argv[1]) : 3000, n = 64 * 1024;
  float* d; CK(hipMalloc(&d, n * 4 * 3)); CK(hipMemset(d, 0, n * 4 * 3));
  hipStream_t sC, sB, sM; CK(hipStreamCreateWithFlags(&sC, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sB, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sM, hipStreamNonBlocking));
  hipEvent_t eC, eB, eM; CK(hipEventCreateWithFlags(&eC, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&eB, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&eM, hipEventDisableTiming));
  Params P; for (int i = 0; i < 200; ++i) P.v[i] = 1.0f;
  const dim3 g(n / 256), b(256);

  auto frameStreams = [&](bool attach) {
    // chain C: 4 kernels, the last carries eC
    hipLaunchKernelGGL(tinyBig, g, b, 0, sC, d, n, P);
    hipLaunchKernelGGL(tiny, g, b, 0, sC, d, n, 1);
    hipLaunchKernelGGL(tiny, g, b, 0, sC, d, n, 2);
    if (attach) hipExtLaunchKernelGGL(tiny, g, b, 0, sC, nullptr, eC, 0, d, n, 3); else { hipLaunchKernelGGL(tiny, g, b, 0, sC, d, n, 3); CK(hipEventRecord(eC, sC)); }
    CK(hipStreamWaitEvent(sB, eC, 0));
    if (attach) hipExtLaunchKernelGGL(tiny, g, b, 0, sB, nullptr, eB, 0, d + n, n, 4); else { hipLaunchKernelGGL(tiny, g, b, 0, sB, d + n, n, 4); CK(hipEventRecord(eB, sB)); }
    CK(hipStreamWaitEvent(sM, eB, 0));
    for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(tiny, g, b, 0, sM, d + 2 * n, n, 5 + k);
    if (attach) hipExtLaunchKernelGGL(tiny, g, b, 0, sM, nullptr, eM, 0, d + 2 * n, n, 9); else { hipLaunchKernelGGL(tiny, g, b, 0, sM, d + 2 * n, n, 9); CK(hipEventRecord(eM, sM)); }
  };
  auto run = [&](const char* name, auto&& body) {
    for (int i = 0; i < 200; ++i) body();
    CK(hipDeviceSynchronize());
    double host = 0.0; const double t0 = now();
    for (int i = 0; i < frames; ++i) { const double a = now(); body(); host += now() - a; }
    CK(hipDeviceSynchronize());
    const double wall = now() - t0;
    printf("%-34s host %.2f us per frame, wall %.2f us per frame\n", name, host / frames * 1e6, wall / frames * 1e6);
  };
  run("streams, events attached", [&] { frameStreams(true); });
  run("streams, hipEventRecord", [&] { frameStreams(false); });
  run("one stream, 10 kernels", [&] { hipLaunchKernelGGL(tinyBig, g, b, 0, sM, d, n, P); for (int k = 1; k < 10; ++k) hipLaunchKernelGGL(tiny, g, b, 0, sM, d, n, k); });

  // the same frame captured: origin stream C, forked to B and M through events, joined back into C
  hipGraph_t graph; hipGraphExec_t exec;
  CK(hipStreamBeginCapture(sC, hipStreamCaptureModeGlobal));
  hipLaunchKernelGGL(tinyBig, g, b, 0, sC, d, n, P);
  hipLaunchKernelGGL(tiny, g, b, 0, sC, d, n, 1);
  hipLaunchKernelGGL(tiny, g, b, 0, sC, d, n, 2);
  hipLaunchKernelGGL(tiny, g, b, 0, sC, d, n, 3);
  CK(hipEventRecord(eC, sC)); CK(hipStreamWaitEvent(sB, eC, 0));
  hipLaunchKernelGGL(tiny, g, b, 0, sB, d + n, n, 4);
  CK(hipEventRecord(eB, sB)); CK(hipStreamWaitEvent(sM, eB, 0));
  for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(tiny, g, b, 0, sM, d + 2 * n, n, 5 + k);
  CK(hipEventRecord(eM, sM)); CK(hipStreamWaitEvent(sC, eM, 0));
  CK(hipStreamEndCapture(sC, &graph));
  CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  size_t numNodes = 0; CK(hipGraphGetNodes(graph, nullptr, &numNodes));
  std::vector<hipGraphNode_t> nodes(numNodes); CK(hipGraphGetNodes(graph, nodes.data(), &numNodes));
  printf("captured graph: %zu nodes\n", numNodes);
  run("graph (3 branches), launch on C", [&] { CK(hipGraphLaunch(exec, sC)); });
  // two graph launches in flight on different streams (the pipeline has 3-4 frames in flight)
  run("graph, alternating streams C / M", [&] { static int k = 0; CK(hipGraphLaunch(exec, (k++ & 1) ? sM : sC)); });

  // replace the arguments of K kernel nodes before every launch
  std::vector<hipGraphNode_t> kn;
  for (auto nd : nodes) { hipGraphNodeType t; CK(hipGraphNodeGetType(nd, &t)); if (t == hipGraphNodeTypeKernel) kn.push_back(nd); }
  for (int K : {1, 3, 10}) {
    if ((size_t)K > kn.size()) continue;
    int tag = 100; float* pp = d; int nn = n;
    char name[64]; snprintf(name, sizeof name, "graph + SetParams of %d nodes", K);
    run(name, [&] {
      for (int k = 0; k < K; ++k) {
        hipKernelNodeParams kp; CK(hipGraphKernelNodeGetParams(kn[k], &kp));
        if (kp.func == (void*)tiny) { ++tag; void* args[3] = {&pp, &nn, &tag}; kp.kernelParams = args; CK(hipGraphExecKernelNodeSetParams(exec, kn[k], &kp)); }
        else { P.v[0] += 1.0f; void* args[3] = {&pp, &nn, &P}; kp.kernelParams = args; CK(hipGraphExecKernelNodeSetParams(exec, kn[k], &kp)); }
      }
      CK(hipGraphLaunch(exec, sC));
    });
  }
  // a linear graph of 10 kernels
  hipGraph_t g2; hipGraphExec_t e2;
  CK(hipStreamBeginCapture(sM, hipStreamCaptureModeGlobal));
  hipLaunchKernelGGL(tinyBig, g, b, 0, sM, d, n, P); for (int k = 1; k < 10; ++k) hipLaunchKernelGGL(tiny, g, b, 0, sM, d, n, k);
  CK(hipStreamEndCapture(sM, &g2)); CK(hipGraphInstantiate(&e2, g2, nullptr, nullptr, 0));
  run("graph, linear 10 kernels", [&] { CK(hipGraphLaunch(e2, sM)); });

  // ---- a frame per stream: the WHOLE frame one linear graph (9 kernels with the durations of a 1920 x 171 strip, 123 us in all),
  // K frames in flight on K streams; what joins consecutive frames -- the history: temporal(f) after temporal(f - 1) -- is an event
  // record node behind node 7 and an event wait node before it (external events, baked per graph: graph k records E[k], waits E[k - 1]).
  const int dur[9] = {13, 9, 10, 47, 11, 8, 10, 10, 5};
  for (int K : {1, 2, 3, 4}) {
    std::vector<hipStream_t> st(K); std::vector<hipEvent_t> ev(K); std::vector<hipGraphExec_t> ex(K);
    for (int k = 0; k < K; ++k) { CK(hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking)); CK(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming)); }
    for (int k = 0; k < K; ++k) {
      hipGraph_t fg; CK(hipGraphCreate(&fg, 0));
      hipGraphNode_t prev = nullptr;
      for (int j = 0; j < 9; ++j) {
        if (j == 7 && K > 1) {      // wait for the previous frame's temporal pass
          hipGraphNode_t w; CK(hipGraphAddEventWaitNode(&w, fg, prev ? &prev : nullptr, prev ? 1 : 0, ev[(k + K - 1) % K])); prev = w;
        }
        float* pp = d + (size_t)k * 64; int ticks = dur[j] * 100, tag = j;
        void* args[3] = {&pp, &ticks, &tag};
        hipKernelNodeParams kp = {}; kp.func = (void*)spin; kp.gridDim = dim3(64); kp.blockDim = dim3(64); kp.kernelParams = args;
        hipGraphNode_t nd; CK(hipGraphAddKernelNode(&nd, fg, prev ? &prev : nullptr, prev ? 1 : 0, &kp)); prev = nd;
        if (j == 7 && K > 1) { hipGraphNode_t r; CK(hipGraphAddEventRecordNode(&r, fg, &prev, 1, ev[k])); prev = r; }
      }
      CK(hipGraphInstantiate(&ex[k], fg, nullptr, nullptr, 0));
    }
    if (K > 1) for (int k = 0; k < K; ++k) CK(hipEventRecord(ev[k], st[k]));     // the first frame's wait finds a recorded event
    char name[64]; snprintf(name, sizeof name, "frame graph per stream, K = %d", K);
    int f = 0;
    run(name, [&] { CK(hipGraphLaunch(ex[f % K], st[f % K])); ++f; });
    // the same frames as individual launches on the K streams (events by hipExtLaunchKernelGGL)
    snprintf(name, sizeof name, "frame per stream, launches, K = %d", K);
    f = 0;
    run(name, [&] {
      const int k = f % K; ++f;
      for (int j = 0; j < 9; ++j) {
        if (j == 7 && K > 1) CK(hipStreamWaitEvent(st[k], ev[(k + K - 1) % K], 0));
        if (j == 7 && K > 1) hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, st[k], nullptr, ev[k], 0, d + (size_t)k * 64, dur[j] * 100, j);
        else hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, st[k], d + (size_t)k * 64, dur[j] * 100, j);
      }
    });
  }
  return 0;
}
