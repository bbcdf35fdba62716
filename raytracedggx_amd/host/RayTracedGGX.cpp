#include <cmath>
#include <cstring>
#include "RayTracedGGX.h"

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <vector>

#define PIDIV4 0.785398163f

static const float g_FOVAngleY = PIDIV4;   // RayTracedGGX.cpp:19-23
static const float g_zNear = 1.0f;
static const float g_zFar = 1000.0f;

RayTracedGGX::RayTracedGGX(uint32_t width, uint32_t height, std::string name) : m_width(width), m_height(height), m_title(std::move(name)) {
  for (auto& metallic : m_metallics) metallic = 1.0f;   // RayTracedGGX.cpp:51
}

RayTracedGGX::~RayTracedGGX() {}

// LoadPipeline + LoadAssets (RayTracedGGX.cpp:61-279)
void RayTracedGGX::OnInit() {
  m_rayTracer = std::make_unique<RayTracer>();
  if (!m_rayTracer->Init(m_width, m_height, m_meshFileName.c_str(), m_envFileName.c_str(), m_meshPosScale, m_device))
    throw std::runtime_error("RayTracer::Init failed: " + m_rayTracer->GetLastError());
  m_denoiser = std::make_unique<Denoiser>();
  if (!m_denoiser->Init(m_rayTracer->GetContext(), m_width, m_height)) throw std::runtime_error("Denoiser::Init failed");
  if (!m_rayTracer->BuildAccelerationStructures()) throw std::runtime_error("BuildAccelerationStructures failed: " + m_rayTracer->GetLastError());
  if (!m_rayTracer->Postinit()) throw std::runtime_error("Postinit failed");
  if (m_hasMetallicOverride) for (uint32_t i = 0; i < RayTracer::NUM_MESH; ++i) m_rayTracer->SetMetallic(i, m_metallics[i]);
  if (m_vndf) m_rayTracer->SetSampler(true);            // -vndf: visible-normal sampling of the reflection lobe (opt-in; the reference samples the NDF)
  m_rayTracer->SetAsyncCompute(m_asyncCompute != 0);   // -sync: one stream, submission order (the sample's single command list)

  if (m_deformAmplitude != 0.0f) {       // key shapes of the breathing model: x and z displaced by a wave travelling up the y axis
    const std::vector<float>& base = m_rayTracer->GetModelVertices();
    m_deformShapes.assign(DeformPeriod, base);
    for (uint32_t k = 0; k < DeformPeriod; ++k) {
      const float phase = 6.283185307f * (float)k / (float)DeformPeriod;
      for (size_t v = 0; v + 5 < base.size(); v += 6) {
        const float y = base[v + 1];
        m_deformShapes[k][v] = base[v] + m_deformAmplitude * std::sin(1.3f * y + phase);
        m_deformShapes[k][v + 2] = base[v + 2] + 0.7f * m_deformAmplitude * std::cos(0.8f * y - phase);
      }
    }
  }
  InitCamera();
  if (!m_trackFileName.empty() && !LoadTrack(m_trackFileName)) throw std::runtime_error("cannot read track " + m_trackFileName);
  m_initialized = true;
}

// Projection and view (RayTracedGGX.cpp:262-277)
void RayTracedGGX::InitCamera() {
  const float aspectRatio = (float)m_width / (float)m_height;
  m_proj = xm::PerspectiveFovLH(g_FOVAngleY, aspectRatio, g_zNear, g_zFar);
  m_focusPt = {0.0f, 3.0f, 0.0f};
  m_eyePt = {10.0f, 10.0f, -24.0f};
  m_view = xm::LookAtLH(m_eyePt, m_focusPt, xm::Float3{0.0f, 1.0f, 0.0f});
}

// RayTracedGGX.cpp:282-299
void RayTracedGGX::OnUpdate() {
  for (; m_trackNext < m_track.size() && m_track[m_trackNext].frame <= m_frameNumber; ++m_trackNext) {
    const TrackEvent& e = m_track[m_trackNext];
    switch (e.type) {
      case 0: OnKeyUp((uint8_t)e.a); break;
      case 1: OnLButtonDown(e.a, e.b); break;
      case 2: OnLButtonUp(e.a, e.b); break;
      case 3: OnMouseMove(e.a, e.b); break;
      case 4: OnMouseWheel(e.a, 0.0f, 0.0f); break;
      default: OnMouseLeave(); break;
    }
  }
  if (!m_deformShapes.empty() && !m_isPaused) {
    const std::vector<float>& shape = m_deformShapes[m_frameNumber % DeformPeriod];
    m_rayTracer->UpdateMesh(shape.data(), (uint32_t)(shape.size() / 6));
  }
  ++m_frameNumber;
  const float timeStep = m_isPaused ? 0.0f : m_fixedTimeStep;
  m_rayTracer->UpdateFrame(m_frameIndex, m_eyePt, m_view * m_proj, timeStep);
}

// RayTracedGGX.cpp:302-353.  asyncCompute: TLAS update on its own stream beside the visibility pass,
// the ray trace waits for it; otherwise everything in submission order (PopulateCommandList :513-556).
void RayTracedGGX::OnRender() {
  m_rayTracer->UpdateAccelerationStructure(m_frameIndex);
  m_rayTracer->RenderVisibility(m_frameIndex, m_asyncCompute != 0);
  m_rayTracer->RayTrace(m_frameIndex);
  m_denoiser->Denoise(m_useSharedMem, m_asyncCompute != 0);
  m_denoiser->ToneMap();
  m_frameIndex = (uint8_t)((m_frameIndex + 1) % FrameCount);   // MoveToNextFrame :684-701
  // Screen-shot helper (MoveToNextFrame :703-717: the sample copies the back buffer of the frame rendered after [F11] and writes
  // "RayTracedGGX_<time stamp>.png" FrameCount frames later).  Headless runs want reproducible names: <-dump prefix or RayTracedGGX>
  // _f<frame number, 6 digits>.png, written at once (the read-back waits for the frame).
  if (m_screenShot) {
    m_screenShot = 0;
    std::string stem = m_dumpPrefix.empty() ? std::string("RayTracedGGX") : m_dumpPrefix;
    if (stem.size() >= 4 && (stem.compare(stem.size() - 4, 4, ".png") == 0 || stem.compare(stem.size() - 4, 4, ".ppm") == 0)) stem.resize(stem.size() - 4);
    char tail[32]; std::snprintf(tail, sizeof tail, "_f%06u.png", m_frameNumber - 1u);      // OnUpdate has counted this frame already
    m_lastScreenShot = stem + tail;
    if (SaveImage(m_lastScreenShot.c_str())) std::printf("wrote %s\n", m_lastScreenShot.c_str()); else m_lastScreenShot.clear();
  }
}

void RayTracedGGX::OnDestroy() {
  if (m_rayTracer && m_rayTracer->GetContext()) rtggx_sync(m_rayTracer->GetContext());   // WaitForGpu
  m_denoiser.reset();
  m_rayTracer.reset();
  m_initialized = false;
}

// RayTracedGGX.cpp:365-398 (key codes: ' ' pause, 0x25/0x27 mesh select, 0x26/0x28 metallic, 'V', 'A')
void RayTracedGGX::OnKeyUp(uint8_t key) {
  float& metallic = m_metallics[m_currentMesh];
  switch (key) {
    case ' ': m_isPaused = !m_isPaused; break;
    case 0x25: m_currentMesh = (m_currentMesh + RayTracer::NUM_MESH - 1) % RayTracer::NUM_MESH; break;
    case 0x27: m_currentMesh = (m_currentMesh + 1) % RayTracer::NUM_MESH; break;
    case 0x26: metallic = std::min(metallic + 0.25f, 1.0f); m_rayTracer->SetMetallic(m_currentMesh, metallic); break;
    case 0x28: metallic = std::max(metallic - 0.25f, 0.0f); m_rayTracer->SetMetallic(m_currentMesh, metallic); break;
    case 0x7A: m_screenShot = 1; break;                       // VK_F11, RayTracedGGX.cpp:388-390: the frame rendered next is saved
    case 'V': m_useSharedMem = !m_useSharedMem; break;
    case 'A': m_asyncCompute = !m_asyncCompute; m_rayTracer->SetAsyncCompute(m_asyncCompute != 0); break;   // RayTracedGGX.cpp:394-396
    default: break;
  }
}

// RayTracedGGX.cpp:400-455: orbit about the focus point while the left button is held, dolly with the wheel
void RayTracedGGX::OnLButtonDown(float posX, float posY) { m_tracking = true; m_mousePt[0] = posX; m_mousePt[1] = posY; }
void RayTracedGGX::OnLButtonUp(float, float) { m_tracking = false; }
void RayTracedGGX::OnMouseLeave() { m_tracking = false; }
static float distance(const xm::Float3& a, const xm::Float3& b) { const xm::Float3 d = xm::Sub(a, b); return std::sqrt(xm::Dot(d, d)); }
void RayTracedGGX::OnMouseMove(float posX, float posY) {
  if (!m_tracking) return;
  const float dx = m_mousePt[0] - posX, dy = m_mousePt[1] - posY;
  const float twoPi = 6.283185307f;                                     // XM_2PI
  const float pitch = twoPi * dy / (float)m_height, yaw = twoPi * dx / (float)m_width;
  const float len = distance(m_focusPt, m_eyePt);
  xm::Matrix transform = xm::Translation(0.0f, 0.0f, -len);
  transform = transform * xm::RotationRollPitchYaw(pitch, yaw, 0.0f);
  transform = transform * xm::Translation(0.0f, 0.0f, len);
  m_view = m_view * transform;
  const xm::Matrix viewInv = xm::Inverse(m_view);
  m_eyePt = {viewInv.r[3][0], viewInv.r[3][1], viewInv.r[3][2]};
  m_mousePt[0] = posX; m_mousePt[1] = posY;
}
void RayTracedGGX::OnMouseWheel(float deltaZ, float, float) {
  const float len = distance(m_focusPt, m_eyePt);
  m_view = m_view * xm::Translation(0.0f, 0.0f, -len * deltaZ / 16.0f);
  const xm::Matrix viewInv = xm::Inverse(m_view);
  m_eyePt = {viewInv.r[3][0], viewInv.r[3][1], viewInv.r[3][2]};
}

bool RayTracedGGX::LoadTrack(const std::string& fileName) {
  FILE* f = std::fopen(fileName.c_str(), "r");
  if (!f) return false;
  m_track.clear(); m_trackNext = 0;
  char line[256];
  while (std::fgets(line, sizeof line, f)) {
    unsigned frame; char cmd[32], arg[32]; float a = 0.0f, b = 0.0f;
    if (line[0] == '#' || std::sscanf(line, "%u %31s", &frame, cmd) != 2) continue;
    const std::string c = cmd;
    TrackEvent e{frame, 5, 0.0f, 0.0f};
    if (c == "key") {
      if (std::sscanf(line, "%u %*s %31s", &frame, arg) != 2) continue;
      const std::string k = arg;
      const int code = k == "SPACE" ? ' ' : k == "LEFT" ? 0x25 : k == "UP" ? 0x26 : k == "RIGHT" ? 0x27 : k == "DOWN" ? 0x28 : k == "F11" ? 0x7A : k.size() == 1 ? std::toupper((unsigned char)k[0]) : std::atoi(arg);
      e.type = 0; e.a = (float)code;
    } else if (c == "down" || c == "up" || c == "move") {
      if (std::sscanf(line, "%u %*s %f %f", &frame, &a, &b) != 3) continue;
      e.type = c == "down" ? 1 : c == "up" ? 2 : 3; e.a = a; e.b = b;
    } else if (c == "wheel") {
      if (std::sscanf(line, "%u %*s %f", &frame, &a) != 2) continue;
      e.type = 4; e.a = a;
    } else if (c != "leave") continue;
    m_track.push_back(e);
  }
  std::fclose(f);
  std::stable_sort(m_track.begin(), m_track.end(), [](const TrackEvent& x, const TrackEvent& y) { return x.frame < y.frame; });
  return true;
}

// RayTracedGGX.cpp:462-511: '-' or '/' prefix, case-insensitive names; a following token is a value
// unless it starts with '/' or with '-' not followed by a digit or '.'.
void RayTracedGGX::ParseCommandLineArgs(char* argv[], int argc) {
  const auto lower = [](std::string s) { std::transform(s.begin(), s.end(), s.begin(), [](unsigned char ch) { return (char)std::tolower(ch); }); return s; };
  const auto isArgMatched = [&](int i, const char* name) {
    const char* arg = argv[i];
    return (arg[0] == '-' || arg[0] == '/') && lower(arg + 1) == lower(name);
  };
  // On POSIX an absolute path also starts with '/': such a token is a flag only when it names one.
  static const char* const kFlags[] = {"warp", "uma", "mesh", "env", "width", "height", "frames", "dt", "metallic", "sharedmem", "sync", "vndf", "device", "dump", "gpus", "track", "deform", "rank", "idfile", "strips", "balance"};
  const auto isFlagName = [&](const char* name) { for (const char* f : kFlags) if (lower(name) == f) return true; return false; };
  const auto hasNextArgValue = [&](int i) {
    if (i + 1 >= argc) return false;
    const char* arg = argv[i + 1];
    if (arg[0] == '/') return !isFlagName(arg + 1);
    return arg[0] != '-' || (arg[1] >= '0' && arg[1] <= '9') || arg[1] == '.';
  };
  const auto nextFloat = [&](int& i, float& dst) { if (hasNextArgValue(i)) { float v; if (std::sscanf(argv[i + 1], "%f", &v) == 1) { dst = v; ++i; } } };
  for (int i = 1; i < argc; ++i) {
    if (isArgMatched(i, "warp") || isArgMatched(i, "uma")) continue;   // device selection of the D3D sample: ignored
    else if (isArgMatched(i, "mesh")) {
      if (hasNextArgValue(i)) m_meshFileName = argv[++i];
      nextFloat(i, m_meshPosScale[0]); nextFloat(i, m_meshPosScale[1]); nextFloat(i, m_meshPosScale[2]); nextFloat(i, m_meshPosScale[3]);
    } else if (isArgMatched(i, "env")) { if (hasNextArgValue(i)) m_envFileName = argv[++i]; }
    // extensions replacing the window / message loop
    else if (isArgMatched(i, "width")) { if (hasNextArgValue(i)) m_width = (uint32_t)std::atoi(argv[++i]); }
    else if (isArgMatched(i, "height")) { if (hasNextArgValue(i)) m_height = (uint32_t)std::atoi(argv[++i]); }
    else if (isArgMatched(i, "frames")) { if (hasNextArgValue(i)) m_numFrames = (uint32_t)std::atoi(argv[++i]); }
    else if (isArgMatched(i, "dt")) { nextFloat(i, m_fixedTimeStep); }
    else if (isArgMatched(i, "metallic")) { nextFloat(i, m_metallics[0]); nextFloat(i, m_metallics[1]); m_hasMetallicOverride = true; }
    else if (isArgMatched(i, "sharedmem")) m_useSharedMem = true;
    else if (isArgMatched(i, "sync")) m_asyncCompute = 0;
    else if (isArgMatched(i, "vndf")) m_vndf = true;
    else if (isArgMatched(i, "device")) { if (hasNextArgValue(i)) m_device = std::atoi(argv[++i]); }
    else if (isArgMatched(i, "dump")) { if (hasNextArgValue(i)) m_dumpPrefix = argv[++i]; }
    else if (isArgMatched(i, "deform")) { nextFloat(i, m_deformAmplitude); }
    else if (isArgMatched(i, "track")) { if (hasNextArgValue(i)) m_trackFileName = argv[++i]; }
    // Several GPUs = one process per GPU, each rendering a strip of rows and exchanging the temporal history over RCCL
    // (host/Strips.cpp; Main.cpp restarts the executable once per rank).  -rank / -idfile are what the launcher passes on;
    // -strips N renders N strips in THIS process on one GPU (the exchange code without N GPUs); -balance 0: equal strips.
    else if (isArgMatched(i, "gpus")) { if (hasNextArgValue(i)) m_gpus = std::atoi(argv[++i]); if (m_gpus < 1 || m_gpus > 64) throw std::runtime_error("-gpus: 1 .. 64"); }
    else if (isArgMatched(i, "rank")) { if (hasNextArgValue(i)) m_rank = std::atoi(argv[++i]); }
    else if (isArgMatched(i, "idfile")) { if (hasNextArgValue(i)) m_idFile = argv[++i]; }
    else if (isArgMatched(i, "strips")) { if (hasNextArgValue(i)) m_strips = std::atoi(argv[++i]); if (m_strips < 1 || m_strips > 64) throw std::runtime_error("-strips: 1 .. 64"); }
    else if (isArgMatched(i, "balance")) { if (hasNextArgValue(i)) m_balance = std::atoi(argv[++i]) != 0; }
  }
}

// PNG, the container the sample's screenshot uses (stbi_write_png, RayTracedGGX.cpp:736): 8-bit RGB or RGBA, one IDAT of
// stored (uncompressed) deflate blocks -- every PNG reader accepts it, and no compression library is needed.
bool WritePng(const char* fileName, uint32_t w, uint32_t h, uint32_t comp, const uint8_t* pixels) {
  if ((comp != 3 && comp != 4) || w == 0 || h == 0) return false;
  static uint32_t crcTable[256];
  if (!crcTable[1]) for (uint32_t n = 0; n < 256; ++n) { uint32_t c = n; for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1; crcTable[n] = c; }
  const auto be32 = [](std::vector<uint8_t>& v, uint32_t x) { v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x); };
  std::vector<uint8_t> file = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
  const auto chunk = [&](const char* type, const std::vector<uint8_t>& data) {
    be32(file, (uint32_t)data.size());
    const size_t start = file.size();
    file.insert(file.end(), type, type + 4); file.insert(file.end(), data.begin(), data.end());
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = start; i < file.size(); ++i) c = crcTable[(c ^ file[i]) & 0xFFu] ^ (c >> 8);
    be32(file, c ^ 0xFFFFFFFFu);
  };
  std::vector<uint8_t> ihdr; be32(ihdr, w); be32(ihdr, h);
  ihdr.push_back(8); ihdr.push_back(comp == 3 ? 2 : 6); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
  chunk("IHDR", ihdr);
  // raw image: per scanline a filter byte (0 = none) + the pixels
  const size_t stride = (size_t)w * comp + 1;
  std::vector<uint8_t> raw(stride * h);
  for (uint32_t y = 0; y < h; ++y) { raw[y * stride] = 0; std::memcpy(&raw[y * stride + 1], pixels + (size_t)y * w * comp, (size_t)w * comp); }
  std::vector<uint8_t> z = {0x78, 0x01};      // zlib header, then stored blocks of at most 65535 bytes
  uint32_t a = 1, b = 0;                       // Adler-32 of the raw data
  for (size_t pos = 0; pos < raw.size();) {
    const size_t n = std::min<size_t>(65535, raw.size() - pos);
    z.push_back(pos + n == raw.size() ? 1 : 0);
    z.push_back((uint8_t)n); z.push_back((uint8_t)(n >> 8)); z.push_back((uint8_t)~n); z.push_back((uint8_t)(~n >> 8));
    z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
    for (size_t i = pos; i < pos + n; ++i) { a = (a + raw[i]) % 65521u; b = (b + a) % 65521u; }
    pos += n;
  }
  be32(z, (b << 16) | a);
  chunk("IDAT", z);
  chunk("IEND", {});
  FILE* f = std::fopen(fileName, "wb");
  if (!f) return false;
  const bool ok = std::fwrite(file.data(), 1, file.size(), f) == file.size();
  std::fclose(f);
  return ok;
}

bool RayTracedGGX::SaveImage(const char* fileName) {
  rtggx_context* ctx = GetContext();
  if (!ctx) return false;
  std::vector<uint32_t> px((size_t)m_width * m_height);
  if (rtggx_readback(ctx, RTGGX_BUF_BACKBUFFER, px.data(), px.size() * 4) != 0) { std::fprintf(stderr, "SaveImage: %s\n", rtggx_last_error()); return false; }
  const std::string name = fileName;
  if (name.size() >= 4 && name.compare(name.size() - 4, 4, ".png") == 0) {      // RGB, as the sample's screenshot (comp = 3)
    std::vector<uint8_t> rgb((size_t)m_width * m_height * 3);
    for (size_t i = 0; i < px.size(); ++i) { rgb[3 * i] = (uint8_t)px[i]; rgb[3 * i + 1] = (uint8_t)(px[i] >> 8); rgb[3 * i + 2] = (uint8_t)(px[i] >> 16); }
    return WritePng(fileName, m_width, m_height, 3, rgb.data());
  }
  FILE* f = std::fopen(fileName, "wb");
  if (!f) return false;
  std::fprintf(f, "P6\n%u %u\n255\n", m_width, m_height);
  std::vector<uint8_t> row((size_t)m_width * 3);
  for (uint32_t y = 0; y < m_height; ++y) {
    for (uint32_t x = 0; x < m_width; ++x) { const uint32_t p = px[(size_t)y * m_width + x]; row[3 * x] = (uint8_t)p; row[3 * x + 1] = (uint8_t)(p >> 8); row[3 * x + 2] = (uint8_t)(p >> 16); }
    std::fwrite(row.data(), 1, row.size(), f);
  }
  std::fclose(f);
  return true;
}
