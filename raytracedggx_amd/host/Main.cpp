// Headless replacement of WinMain + Win32Application::Run (RayTracedGGX/Main.cpp:15-20,
// Common/Win32Application.cpp:17-79, 205-211): one OnUpdate + OnRender per frame, no window.
// The argument tail of each Bin/*.bat works unchanged, e.g.
//   RayTracedGGX -mesh Assets/bunny.obj 0.0 0.0 0.0 1.0 -width 1920 -height 1080 -frames 64 -dump out
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include "RayTracedGGX.h"
#include "Strips.h"

int main(int argc, char* argv[]) {
  // five streams with the strip exchange of -gpus N (rtggx.h rtggx_get_exchange_stream): HIP's default is four hardware queues.  Read when
  // the runtime initialises, which nothing has made it do yet; the ranks the launcher starts inherit it.
  setenv("GPU_MAX_HW_QUEUES", "8", 0);
  RayTracedGGX app(1280, 720, "DXR Ray-Traced GGX");   // Main.cpp:17
  try {
    app.ParseCommandLineArgs(argv, argc);
    // several GPUs: one process per GPU (nothing has touched a GPU yet: the launcher may restart the executable), each a strip of rows
    if (app.GetNumGpus() > 1 && app.GetRank() < 0) return strips::LaunchRanks(app.GetNumGpus(), argc, argv);
    if (app.GetNumGpus() > 1) return strips::RunRank(app, app.GetRank(), app.GetNumGpus(), app.GetIdFile(), app.GetBalance());
    if (app.GetNumStrips() > 1) return strips::RunStripsInOneProcess(app, app.GetNumStrips(), app.GetBalance(), argc, argv);
    app.OnInit();
    rtggx_context* ctx = app.GetContext();
    const auto t0 = std::chrono::steady_clock::now();
    uint64_t rays = 0;
    for (uint32_t f = 0; f < app.GetNumFrames(); ++f) { app.OnUpdate(); app.OnRender(); }
    rtggx_ray_count(ctx, &rays);   // synchronises
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::printf("%u frames %ux%u: %.3f ms/frame, last frame %llu rays\n", app.GetNumFrames(), app.GetWidth(), app.GetHeight(),
                ms / app.GetNumFrames(), (unsigned long long)rays);
    if (!app.GetDumpPrefix().empty()) {
      std::string name = app.GetDumpPrefix();      // "-dump shot.png" / "-dump shot.ppm" as given, a bare prefix gets .ppm
      const bool hasExt = name.size() >= 4 && (name.compare(name.size() - 4, 4, ".png") == 0 || name.compare(name.size() - 4, 4, ".ppm") == 0);
      if (!hasExt) name += ".ppm";
      if (app.SaveImage(name.c_str())) std::printf("wrote %s\n", name.c_str());
    }
    app.OnDestroy();
  } catch (const std::exception& e) {
    std::fprintf(stderr, "RayTracedGGX: %s\n", e.what());
    return 1;
  }
  return 0;
}
