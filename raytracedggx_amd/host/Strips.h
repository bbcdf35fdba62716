// Multi-GPU host (BASELINE.json north_star: "RCCL under the C++ host"): screen-space row strips, one process per GPU.
//
// Rank r of N renders rows [b_r, e_r) of the frame (rtggx_set_strip: the passes that are pure functions of the pixel recompute the
// apron rows they need).  The one exchange per frame is the temporal history -- the HistoryApron rows of TemporalSSOut[parity] on
// each side of a strip boundary go to the neighbour -- plus the gather of the tone-mapped strips on rank 0 (the reference presents
// one back buffer, RayTracedGGX.cpp:341-353).  Point-to-point ncclSend / ncclRecv in one group per frame on the context's main
// stream (rtggx_get_stream): ordered behind the frame's tone map, needed by the next frame's temporal pass only.  No collective.
// Every rank also maps every other rank's two history images (hipIpc through files beside the ncclUniqueId; plain pointers in the
// single-process mode): a history tap beyond the exchanged apron reads the owner's image, so N strips equal the single-GPU frame at any velocity.
// This is the C++ twin of raytracedggx_amd/strips.py (what bench.py drives); the plan functions give the same answers
// (tests/test_gpu_parity.py runs the executable's single-process mode against the single-context frame).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

class RayTracedGGX;

namespace strips {

constexpr uint32_t HistoryApron = 18;     // rows: 1 (the temporal pass also computes rows b-1 and e for the tone map) + 16 px/frame of vertical reprojection + 1 (bilinear footprint)
constexpr uint32_t ProfileFrames = 2;     // whole frames every rank renders first when it balances the strips itself
constexpr double SkyRowWeight = 0.5;      // cost of a row = covered pixels + this x width (strips.py SKY_ROW_WEIGHT)
constexpr double GatherRowWeight = 0.02;  // what a row rank 0 RECEIVES in the gather costs it, x width (strips.py GATHER_ROW_WEIGHT)

// Rows [begin, end) of `rank`: equal strips, or `bounds` (world + 1 ascending row numbers from 0 to height).
std::pair<uint32_t, uint32_t> StripRows(uint32_t height, int rank, int world, const std::vector<uint32_t>* bounds = nullptr);
// Boundaries that even out sum(rowCost) per strip; every strip at least minRows rows.  firstExtra: a cost rank 0 carries besides its rows
// (the gather: GatherCost), so its strip ends where its rows cost that much less than the others'.  Deterministic: every rank computes the same.
std::vector<uint32_t> BalancedBounds(const std::vector<double>& rowCost, int world, uint32_t minRows = HistoryApron, double firstExtra = 0.0);
double GatherCost(uint32_t width, uint32_t height, int world, double rowWeight = GatherRowWeight);

enum class Buffer { History /* TemporalSSOut[parity], 8 B/px */, BackBuffer /* 4 B/px */, Token /* words of RTGGX_BUF_EXCHANGE_TOKENS */ };
struct Op { bool send; Buffer buffer; uint32_t rowBegin, rowEnd; int peer; };      // rows of the frame; for a token: words of the token buffer
// The transfers of one frame for `rank`.  Ops between a pair of ranks appear in the same order on both sides (history first,
// then the back-buffer strip, then the token): that is what tag-less send / recv matching needs.
// tokens (round 4): every rank maps every rank's history images (rtggx_set_history_peers: a history tap beyond the apron reads the
// owner's image), which needs an order between ANY two ranks, in both directions, every frame; a message each way in this exchange gives
// it.  Neighbours have their history rows, everybody has a back-buffer strip for rank 0; the pairs and directions that have nothing get
// a 4-byte token (word [rank] of the token buffer to send from, [RTGGX_MAX_PEERS + peer] to receive into).
std::vector<Op> ExchangePlan(uint32_t height, int rank, int world, uint32_t apron = HistoryApron, const std::vector<uint32_t>* bounds = nullptr, bool tokens = true);

struct RawOp { bool send; void* ptr; size_t bytes; int peer; };
// librccl.so, loaded at run time (the single-GPU executable does not need it).
class Rccl {
 public:
  Rccl();                                      // throws std::runtime_error when the library or a symbol is missing
  ~Rccl();
  // One communicator over `world` ranks.  idFile: rank 0 writes the ncclUniqueId there (atomically), the others wait for it;
  // empty with world == 1 (single-process mode: sends and receives pair up inside one rank).
  void InitRank(int rank, int world, const std::string& idFile);
  void Exchange(const std::vector<RawOp>& ops, void* hipStream);      // one ncclGroupStart / End
 private:
  void* m_lib = nullptr; void* m_comm = nullptr;
  void* m_fn[8] = {};
};

// `plan` over this application's TemporalSSOut[parity] and back buffer: every rank holds full-size targets, so both sides of a
// transfer address the same rows.
std::vector<RawOp> PlanToRaw(const std::vector<Op>& plan, RayTracedGGX& app);

// Renders ProfileFrames whole frames and cuts the frame where covered pixels (+ SkyRowWeight x width per row) balance.
std::vector<uint32_t> ProfileBounds(RayTracedGGX& app, int world, uint32_t apron = HistoryApron);

// One process per GPU: the executable restarts itself `world` times with -rank r -idfile <path> (before anything touches a GPU)
// and returns the worst exit code.
// Restarts the executable once per rank (before anything has touched a GPU: it refuses, exit code 2, under a profiler / preload or with
// /dev/kfd open), reaps the ranks in any order, and stops the others when one fails or none has ended for RankTimeoutSeconds
// (RTGGX_RANK_TIMEOUT overrides).  The ranks, not the launcher, are what belongs behind `rocprofv3 --`.
constexpr double RankTimeoutSeconds = 900.0, RankGraceSeconds = 2.0;
int LaunchRanks(int world, int argc, char* argv[]);
// Body of a rank (world > 1), and of the single-process mode (-strips N: N contexts on one GPU, the sends and receives of a
// one-rank communicator paired with each other -- the exchange code exercised without N GPUs).  Return the process exit code.
int RunRank(RayTracedGGX& app, int rank, int world, const std::string& idFile, bool balance);
int RunStripsInOneProcess(RayTracedGGX& app, int strips, bool balance, int argc, char* argv[]);

}  // namespace strips
