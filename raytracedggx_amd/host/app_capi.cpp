// C entry points over the C++ host classes, for drivers that are not C++ (bench.py, tests):
// the frame entry stays RayTracedGGX::OnUpdate / OnRender.
#include <cstring>
#include <exception>
#include <string>
#include <vector>
#include "ObjLoader.h"
#include "RayTracedGGX.h"
#include "Strips.h"

static thread_local std::string g_appError;

extern "C" {

const char* rtggx_app_last_error(void) { return g_appError.c_str(); }

// argv-style construction: the same flags as the RayTracedGGX executable.
void* rtggx_app_create(int argc, char** argv) {
  try {
    RayTracedGGX* app = new RayTracedGGX(1280, 720, "DXR Ray-Traced GGX");
    app->ParseCommandLineArgs(argv, argc);
    app->OnInit();
    return app;
  } catch (const std::exception& e) { g_appError = e.what(); return nullptr; }
}
void rtggx_app_destroy(void* h) { RayTracedGGX* app = (RayTracedGGX*)h; if (app) { app->OnDestroy(); delete app; } }
void rtggx_app_on_update(void* h) { ((RayTracedGGX*)h)->OnUpdate(); }
void rtggx_app_on_render(void* h) { ((RayTracedGGX*)h)->OnRender(); }
void rtggx_app_on_key_up(void* h, int key) { ((RayTracedGGX*)h)->OnKeyUp((uint8_t)key); }
void rtggx_app_on_lbutton_down(void* h, float x, float y) { ((RayTracedGGX*)h)->OnLButtonDown(x, y); }
void rtggx_app_on_lbutton_up(void* h, float x, float y) { ((RayTracedGGX*)h)->OnLButtonUp(x, y); }
void rtggx_app_on_mouse_move(void* h, float x, float y) { ((RayTracedGGX*)h)->OnMouseMove(x, y); }
void rtggx_app_on_mouse_wheel(void* h, float dz, float x, float y) { ((RayTracedGGX*)h)->OnMouseWheel(dz, x, y); }
int rtggx_app_load_track(void* h, const char* path) { return ((RayTracedGGX*)h)->LoadTrack(path) ? 0 : -1; }
// The camera handlers on their own (no device): events = n x (type, a, b) with type 1 down, 2 up, 3 move, 4 wheel;
// returns the eye point and the view matrix (row-major) they lead to from the sample's initial camera.
void rtggx_host_camera(uint32_t width, uint32_t height, const float* events, uint32_t n, float* eye3, float* view16) {
  RayTracedGGX app(width, height, "");
  app.InitCamera();
  for (uint32_t i = 0; i < n; ++i) {
    const float* e = events + 3 * i;
    switch ((int)e[0]) { case 1: app.OnLButtonDown(e[1], e[2]); break; case 2: app.OnLButtonUp(e[1], e[2]); break;
                         case 3: app.OnMouseMove(e[1], e[2]); break; case 4: app.OnMouseWheel(e[1], 0.0f, 0.0f); break; default: app.OnMouseLeave(); }
  }
  eye3[0] = app.GetEyePt().x; eye3[1] = app.GetEyePt().y; eye3[2] = app.GetEyePt().z;
  std::memcpy(view16, app.GetView().r, 64);
}
void rtggx_app_set_time_step(void* h, float dt) { ((RayTracedGGX*)h)->SetFixedTimeStep(dt); }
void* rtggx_app_context(void* h) { return ((RayTracedGGX*)h)->GetContext(); }
void rtggx_app_size(void* h, uint32_t* w, uint32_t* ht) { *w = ((RayTracedGGX*)h)->GetWidth(); *ht = ((RayTracedGGX*)h)->GetHeight(); }
void rtggx_app_frame_constants(void* h, void* out768) { std::memcpy(out768, &((RayTracedGGX*)h)->GetRayTracer()->GetFrameConstants(), sizeof(RtggxFrameConstants)); }
int rtggx_app_set_dump_prefix(void* h, const char* prefix) { ((RayTracedGGX*)h)->SetDumpPrefix(prefix ? prefix : ""); return 0; }
const char* rtggx_app_last_screen_shot(void* h) { return ((RayTracedGGX*)h)->GetLastScreenShot().c_str(); }
int rtggx_app_save_image(void* h, const char* path) { return ((RayTracedGGX*)h)->SaveImage(path) ? 0 : -1; }

int rtggx_host_write_png(const char* path, uint32_t w, uint32_t h, uint32_t comp, const uint8_t* pixels) { return WritePng(path, w, h, comp, pixels) ? 0 : -1; }

// Host-only pieces, usable without a GPU: the OBJ importer and the Halton sequence.
static ObjLoader g_obj;
int rtggx_host_obj_import(const char* path, uint32_t* numVerts, uint32_t* numIndices, float* aabb6) {
  g_obj = ObjLoader();
  if (!g_obj.Import(path, true, true)) { g_appError = std::string("cannot import ") + path; return -1; }
  *numVerts = g_obj.GetNumVertices(); *numIndices = g_obj.GetNumIndices();
  if (aabb6) { const ObjLoader::AABB& a = g_obj.GetAABB(); aabb6[0] = a.Min.x; aabb6[1] = a.Min.y; aabb6[2] = a.Min.z; aabb6[3] = a.Max.x; aabb6[4] = a.Max.y; aabb6[5] = a.Max.z; }
  return 0;
}
void rtggx_host_obj_copy(float* verts, uint32_t* indices) {
  std::memcpy(verts, g_obj.GetVertices(), (size_t)g_obj.GetNumVertices() * 24);
  std::memcpy(indices, g_obj.GetIndices(), (size_t)g_obj.GetNumIndices() * 4);
}
void rtggx_host_halton(uint32_t n, float* xy) { HaltonSequence h; for (uint32_t i = 0; i < n; ++i) h.Next(xy[2 * i], xy[2 * i + 1]); }

// RayTracer::UpdateFrame on its own (constants only, no device): `frames` consecutive frames with a
// fixed time step and the default camera; writes frames x 768 bytes.
class ConstantsOnlyRayTracer : public RayTracer {
 public:
  void Setup(uint32_t w, uint32_t h, const float* ps) { m_width = w; m_height = h; std::memcpy(m_posScale, ps, 16); }
};
}  // extern "C"

// UpdateFrame calls rtggx_update_frame; with a null context that call only reports an error, the constants are still produced.
extern "C" void rtggx_host_frame_constants(uint32_t width, uint32_t height, const float* posScale4, const float* eye3, const float* focus3,
                                           float dt, uint32_t frames, void* out) {
  ConstantsOnlyRayTracer rt; rt.Setup(width, height, posScale4);
  const xm::Float3 eye{eye3[0], eye3[1], eye3[2]}, focus{focus3[0], focus3[1], focus3[2]};
  const xm::Matrix proj = xm::PerspectiveFovLH(0.785398163f, (float)width / (float)height, 1.0f, 1000.0f);
  const xm::Matrix view = xm::LookAtLH(eye, focus, xm::Float3{0.0f, 1.0f, 0.0f});
  for (uint32_t f = 0; f < frames; ++f) {
    rt.UpdateFrame((uint8_t)(f % 3), eye, view * proj, dt);
    std::memcpy((char*)out + 768 * (size_t)f, &rt.GetFrameConstants(), 768);
  }
}

// The multi-GPU host's plan functions (host/Strips.cpp), for the tests that compare them with raytracedggx_amd/strips.py.
// bounds: nullptr / 0 for equal strips, else world + 1 row numbers.  ops: 5 int32 per transfer (send, buffer: 1 history / 0 back buffer / 2 token, rowBegin, rowEnd, peer);
// returns the number of transfers, or -1 with rtggx_app_last_error set.
extern "C" int rtggx_host_exchange_plan(uint32_t height, int rank, int world, uint32_t apron, const uint32_t* bounds, int32_t* ops, int capacity) {
  try {
    const std::vector<uint32_t> b(bounds, bounds + (bounds ? world + 1 : 0));
    const std::vector<strips::Op> plan = strips::ExchangePlan(height, rank, world, apron, &b);
    if ((int)plan.size() > capacity) { g_appError = "rtggx_host_exchange_plan: capacity"; return -1; }
    for (size_t i = 0; i < plan.size(); ++i) { int32_t* o = ops + 5 * i; o[0] = plan[i].send; o[1] = plan[i].buffer == strips::Buffer::History ? 1 : plan[i].buffer == strips::Buffer::BackBuffer ? 0 : 2; o[2] = (int32_t)plan[i].rowBegin; o[3] = (int32_t)plan[i].rowEnd; o[4] = plan[i].peer; }
    return (int)plan.size();
  } catch (const std::exception& e) { g_appError = e.what(); return -1; }
}
extern "C" int rtggx_host_balanced_bounds(const double* rowCost, uint32_t height, int world, uint32_t minRows, double firstExtra, uint32_t* boundsOut) {
  try {
    const std::vector<uint32_t> b = strips::BalancedBounds(std::vector<double>(rowCost, rowCost + height), world, minRows, firstExtra);
    for (size_t i = 0; i < b.size(); ++i) boundsOut[i] = b[i];
    return 0;
  } catch (const std::exception& e) { g_appError = e.what(); return -1; }
}
