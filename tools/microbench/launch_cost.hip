// Host cost of the HIP calls a frame of librtggx is made of, one at a time, and of the candidate ways to issue the main stream's chain
// of five kernels (profiles/r03_a_launch_cost.txt).  Every figure: wall clock around N calls, free-running, one synchronise at the end.
// Build: hipcc -O2 --offload-arch=gfx950 launch_cost.hip -o launch_cost
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
struct Params { float v[228]; };      // 912 bytes by value, like FrameParams
__global__ void tiny(float* p, int n, int tag) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + (float)tag; }
__global__ void tinyBig(float* p, int n, Params q) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += q.v[i % 228]; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 4000, n = 16 * 1024;
  float* d; CK(hipMalloc(&d, n * 4 * 4)); CK(hipMemset(d, 0, n * 4 * 4));
  hipStream_t sA, sB, sC; CK(hipStreamCreateWithFlags(&sA, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sB, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sC, hipStreamNonBlocking));
  hipEvent_t eA, eB, eC; CK(hipEventCreateWithFlags(&eA, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&eB, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&eC, hipEventDisableTiming));
  Params P; for (float& v : P.v) v = 1.0f;
  const dim3 g(n / 256), b(256);
  auto run = [&](const char* name, auto&& body) {
    for (int i = 0; i < 100; ++i) body();
    CK(hipDeviceSynchronize());
    const double t0 = now();
    for (int i = 0; i < N; ++i) body();
    const double host = now() - t0;
    CK(hipDeviceSynchronize());
    const double wall = now() - t0;
    printf("%-58s host %6.2f us   wall %6.2f us\n", name, host / N * 1e6, wall / N * 1e6);
  };
  run("hipLaunchKernelGGL, 3 small args", [&] { hipLaunchKernelGGL(tiny, g, b, 0, sA, d, n, 1); });
  run("hipLaunchKernelGGL, 912-byte argument", [&] { hipLaunchKernelGGL(tinyBig, g, b, 0, sA, d, n, P); });
  run("hipExtLaunchKernelGGL + stop event", [&] { hipExtLaunchKernelGGL(tiny, g, b, 0, sA, nullptr, eA, 0, d, n, 1); });
  run("hipLaunchKernelGGL + hipEventRecord", [&] { hipLaunchKernelGGL(tiny, g, b, 0, sA, d, n, 1); CK(hipEventRecord(eA, sA)); });
  run("hipEventRecord alone", [&] { CK(hipEventRecord(eA, sA)); });
  run("hipStreamWaitEvent (event of another stream, complete)", [&] { CK(hipStreamWaitEvent(sB, eA, 0)); });
  run("hipEventQuery", [&] { (void)hipEventQuery(eA); });
  run("kernel A(ext,ev) -> wait -> kernel B (two streams)", [&] { hipExtLaunchKernelGGL(tiny, g, b, 0, sA, nullptr, eA, 0, d, n, 1); CK(hipStreamWaitEvent(sB, eA, 0)); hipLaunchKernelGGL(tiny, g, b, 0, sB, d + n, n, 2); });
  run("5 kernels, one stream", [&] { for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(tiny, g, b, 0, sA, d, n, k); });
  run("4 kernels + 1 (ext, event), one stream", [&] { for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(tiny, g, b, 0, sA, d, n, k); hipExtLaunchKernelGGL(tiny, g, b, 0, sA, nullptr, eA, 0, d, n, 4); });

  // linear graphs of K kernels
  for (int K : {1, 2, 3, 5, 8}) {
    hipGraph_t gr; hipGraphExec_t ex;
    CK(hipStreamBeginCapture(sA, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < K; ++k) hipLaunchKernelGGL(tiny, g, b, 0, sA, d, n, k);
    CK(hipStreamEndCapture(sA, &gr)); CK(hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0));
    char name[96]; snprintf(name, sizeof name, "graph of %d kernels (linear), hipGraphLaunch", K);
    run(name, [&] { CK(hipGraphLaunch(ex, sA)); });
    if (K == 5) {
      size_t nn = 0; CK(hipGraphGetNodes(gr, nullptr, &nn)); std::vector<hipGraphNode_t> nodes(nn); CK(hipGraphGetNodes(gr, nodes.data(), &nn));
      int tag = 7; float* pp = d; int cnt = n;
      run("graph of 5 kernels + SetParams of all 5 (args and grid)", [&] {
        for (size_t k = 0; k < nn; ++k) {
          hipKernelNodeParams kp{}; kp.func = (void*)tiny; kp.gridDim = dim3(n / 256 - (tag & 1)); kp.blockDim = b; kp.sharedMemBytes = 0; ++tag;
          void* args[3] = {&pp, &cnt, &tag}; kp.kernelParams = args; kp.extra = nullptr;
          CK(hipGraphExecKernelNodeSetParams(ex, nodes[k], &kp));
        }
        CK(hipGraphLaunch(ex, sA));
      });
      run("graph of 5 kernels + hipEventRecord behind it", [&] { CK(hipGraphLaunch(ex, sA)); CK(hipEventRecord(eA, sA)); });
    }
  }
  // a graph with an external event record node as its last node
  {
    hipGraph_t gr; hipGraphExec_t ex;
    CK(hipStreamBeginCapture(sA, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(tiny, g, b, 0, sA, d, n, k);
    hipError_t e = hipEventRecordWithFlags(eC, sA, hipEventRecordExternal);
    hipError_t e2 = hipStreamEndCapture(sA, &gr);
    if (e == hipSuccess && e2 == hipSuccess && hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0) == hipSuccess) {
      run("graph of 5 kernels + external event record node", [&] { CK(hipGraphLaunch(ex, sA)); });
      run("  ... and another stream waits for that event + 1 kernel", [&] { CK(hipGraphLaunch(ex, sA)); CK(hipStreamWaitEvent(sB, eC, 0)); hipLaunchKernelGGL(tiny, g, b, 0, sB, d + n, n, 2); });
    } else printf("external event record node: not supported here (%s / %s)\n", hipGetErrorString(e), hipGetErrorString(e2));
  }
  // can a kernel launched with hipExtLaunchKernelGGL + event be captured?
  {
    hipGraph_t gr = nullptr;
    (void)hipGetLastError();
    CK(hipStreamBeginCapture(sA, hipStreamCaptureModeThreadLocal));
    hipExtLaunchKernelGGL(tiny, g, b, 0, sA, nullptr, eA, 0, d, n, 1);
    hipError_t e = hipGetLastError();
    hipError_t e2 = hipStreamEndCapture(sA, &gr);
    printf("capture of hipExtLaunchKernelGGL with a stop event: launch %s, end capture %s\n", hipGetErrorString(e), hipGetErrorString(e2));
  }
  // the pipeline's shape with the main chain as a graph: C: 2 kernels (2nd carries eC) | B: wait, 1 kernel (carries eB) | A: wait, graph of 5
  {
    hipGraph_t gr; hipGraphExec_t ex;
    CK(hipStreamBeginCapture(sA, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(tiny, g, b, 0, sA, d, n, k);
    CK(hipStreamEndCapture(sA, &gr)); CK(hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0));
    run("frame: C 2 kernels | B 1 kernel | main = graph of 5", [&] {
      hipLaunchKernelGGL(tinyBig, g, b, 0, sC, d + 2 * n, n, P); hipExtLaunchKernelGGL(tiny, g, b, 0, sC, nullptr, eC, 0, d + 2 * n, n, 1);
      CK(hipStreamWaitEvent(sB, eC, 0)); hipExtLaunchKernelGGL(tiny, g, b, 0, sB, nullptr, eB, 0, d + n, n, 2);
      CK(hipStreamWaitEvent(sA, eB, 0)); CK(hipGraphLaunch(ex, sA));
    });
    run("frame: C 2 kernels | B 1 kernel | main = 5 kernels", [&] {
      hipLaunchKernelGGL(tinyBig, g, b, 0, sC, d + 2 * n, n, P); hipExtLaunchKernelGGL(tiny, g, b, 0, sC, nullptr, eC, 0, d + 2 * n, n, 1);
      CK(hipStreamWaitEvent(sB, eC, 0)); hipExtLaunchKernelGGL(tiny, g, b, 0, sB, nullptr, eB, 0, d + n, n, 2);
      CK(hipStreamWaitEvent(sA, eB, 0)); for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(tiny, g, b, 0, sA, d, n, k);
    });
  }
  return 0;
}
