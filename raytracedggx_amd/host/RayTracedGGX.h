// Headless frame driver with the reference's entry points (RayTracedGGX/RayTracedGGX.h:27-146,
// DXFramework virtuals RayTracedGGX/Common/DXFramework.h:23-26): OnInit / OnUpdate / OnRender /
// OnDestroy, the same command line (RayTracedGGX.cpp:462-511) and the same defaults
// (RayTracedGGX.cpp:37-39, camera :19-23, 267-277).  What the window supplied interactively is
// supplied by extra flags: -width -height -frames -dt -metallic -sharedmem -sync -device -dump -track -deform -gpus -strips -balance.
#pragma once
#include <vector>
#include <cstdint>
#include <memory>
#include <string>
#include "Denoiser.h"
#include "RayTracer.h"
#include "XMath.h"

bool WritePng(const char* fileName, uint32_t w, uint32_t h, uint32_t comp, const uint8_t* pixels);   // comp 3 (RGB) or 4 (RGBA), 8 bits

class RayTracedGGX {
 public:
  RayTracedGGX(uint32_t width, uint32_t height, std::string name);
  virtual ~RayTracedGGX();

  virtual void OnInit();
  virtual void OnUpdate();
  virtual void OnRender();
  virtual void OnDestroy();
  virtual void OnKeyUp(uint8_t key);
  // camera interactions of the sample's window (RayTracedGGX.h:58-63, .cpp:400-455): positions in pixels
  virtual void OnLButtonDown(float posX, float posY);
  virtual void OnLButtonUp(float posX, float posY);
  virtual void OnMouseMove(float posX, float posY);
  virtual void OnMouseWheel(float deltaZ, float posX, float posY);
  virtual void OnMouseLeave();
  void InitCamera();                                 // projection and view of LoadAssets (RayTracedGGX.cpp:262-277)
  const xm::Float3& GetEyePt() const { return m_eyePt; }
  const xm::Matrix& GetView() const { return m_view; }
  // Scripted input (replaces the message loop): a text file, one event per line, applied by OnUpdate before the frame
  // with that number:  <frame> key <code | SPACE LEFT RIGHT UP DOWN V A> | down x y | up x y | move x y | wheel dz | leave
  bool LoadTrack(const std::string& fileName);

  void ParseCommandLineArgs(char* argv[], int argc);

  bool IsInitialized() const { return m_initialized; }
  uint32_t GetWidth() const { return m_width; }
  uint32_t GetHeight() const { return m_height; }
  uint32_t GetNumFrames() const { return m_numFrames; }
  const std::string& GetDumpPrefix() const { return m_dumpPrefix; }
  void SetDumpPrefix(const std::string& prefix) { m_dumpPrefix = prefix; }
  const std::string& GetLastScreenShot() const { return m_lastScreenShot; }
  // multi-GPU (host/Strips.h): -gpus N (one process per GPU), what the launcher hands a rank, the single-process mode
  int GetNumGpus() const { return m_gpus; }
  int GetRank() const { return m_rank; }
  const std::string& GetIdFile() const { return m_idFile; }
  int GetNumStrips() const { return m_strips; }
  bool GetBalance() const { return m_balance; }
  RayTracer* GetRayTracer() const { return m_rayTracer.get(); }
  rtggx_context* GetContext() const { return m_rayTracer ? m_rayTracer->GetContext() : nullptr; }
  void SetFixedTimeStep(float dt) { m_fixedTimeStep = dt; }
  bool SaveImage(const char* fileName);   // tone-mapped back buffer as PNG (name ends in .png) or binary PPM (screenshot, RayTracedGGX.cpp:719-739)

 protected:
  static const uint8_t FrameCount = RayTracer::FrameCount;

  uint32_t m_width, m_height;
  std::string m_title;
  std::unique_ptr<RayTracer> m_rayTracer;
  std::unique_ptr<Denoiser> m_denoiser;
  uint8_t m_frameIndex = 0;
  bool m_initialized = false;

  // toggles of the reference (RayTracedGGX.h:118-124)
  int m_asyncCompute = 1;
  uint32_t m_currentMesh = 0;
  bool m_useSharedMem = false;
  bool m_isPaused = false;
  uint32_t m_screenShot = 0;           // RayTracedGGX.h:123; set by [F11]
  std::string m_lastScreenShot;        // the file the last [F11] wrote
  float m_metallics[RayTracer::NUM_MESH];

  // camera (RayTracedGGX.h:100-104)
  xm::Matrix m_proj, m_view;
  xm::Float3 m_focusPt, m_eyePt;
  bool m_tracking = false;
  float m_mousePt[2] = {0.0f, 0.0f};
  struct TrackEvent { uint32_t frame; int type; float a, b; };   // type: 0 key, 1 down, 2 up, 3 move, 4 wheel, 5 leave
  std::vector<TrackEvent> m_track;
  size_t m_trackNext = 0;
  uint32_t m_frameNumber = 0;
  std::string m_trackFileName;

  // command line
  std::string m_meshFileName = "Assets/dragon.obj";
  std::string m_envFileName = "Assets/rnl_cross.dds";
  float m_meshPosScale[4] = {0.0f, 0.0f, 0.0f, 1.0f};
  uint32_t m_numFrames = 1;
  float m_fixedTimeStep = 1.0f / 60.0f;   // the reference steps by the wall clock (StepTimer); fixed here for reproducible runs
  int m_device = 0;
  std::string m_dumpPrefix;
  int m_gpus = 1, m_rank = -1, m_strips = 1; bool m_balance = true; std::string m_idFile;
  bool m_hasMetallicOverride = false;
  bool m_vndf = false;                 // -vndf
  // -deform <amplitude>: the model breathes -- a travelling sine wave through its vertices, DeformPeriod key shapes computed once
  // at start-up and handed to RayTracer::UpdateMesh one per frame (per-frame host cost: one copy of the vertex array)
  float m_deformAmplitude = 0.0f;
  static const uint32_t DeformPeriod = 32;
  std::vector<std::vector<float>> m_deformShapes;
};
