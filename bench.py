#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X.

  metric   : Mrays/s (+ ms/frame in `ms_per_step`), bunny + rnl env, 1920x1080, 1spp + full denoise chain
  step     : one frame = RayTracedGGX::OnUpdate + OnRender (visibility -> ray trace -> 4 spatial passes -> temporal -> tone map)
  N GPUs   : the SAME 1920x1080 frame sharded by row strips, one process per GPU, history-apron exchange over RCCL + frame
             gather on rank 0 ("scaling": "strong").  Started by a launcher (torch.distributed.run sets RANK / LOCAL_RANK /
             WORLD_SIZE / MASTER_*) this process is one rank; started plainly with --gpus N > 1 it becomes a LAUNCHER: it
             starts N fresh child processes (one rank each) before anything here has touched the GPU, relays rank 0's JSON
             line and exits with the worst child's code.
  value    : non-degenerate rays traced by all ranks in the K timed frames / max-over-ranks wall time
  roofline : the dominant kernel (rt::traceKernel: BVH traversal of the binned rays), algorithmic bytes per launch
             (DESIGN.md "Roofline accounting") / its average duration from HIP events recorded on the launching stream
             (stream B); `traffic` = HBM-side bytes per launch from the newest profiles/*_pmc_traffic.json;
             `peak_measured` = a float4 copy kernel on this box; `roofline.frame` = SURVEY 8(d)'s whole-frame bytes / frame time
  cpu_baseline : the scalar C++ oracle re-tracing the SAME BVH arrays on the host cores, bounded sample (rank 0, N=1 only):
             all threads, and one thread

Usage: python bench.py [--gpus N] [--steps K] [--warmup W] [--width 1920 --height 1080 --mesh bunny.obj]
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

# The renderer runs a frame on four streams, RCCL brings its own, the null stream is a fifth: HIP's default of four hardware queues would
# make two of them share one.  Read when the runtime initialises, i.e. before torch is imported; the ranks inherit it.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
VALU_CYCLES_PER_WAVE_INSTRUCTION = 2.0   # v_fma_f32 wave64 on a SIMD-32 with >= 2 waves resident (same guide; tools/microbench/valu_issue.hip measures it)
SHADER_CLOCK_HZ = 2.4e9


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--mesh", default="bunny.obj")
    ap.add_argument("--metallic", type=float, nargs=2, default=None, help="metallic of ground and model (default: the sample's 1 1 = no diffuse rays)")
    ap.add_argument("--deform", type=float, default=0.0, help="amplitude of the breathing-model animation (-deform): new vertices and an asynchronous BVH refit every frame")
    ap.add_argument("--trace-waves", type=int, default=0, help="pin the size of the traversal's resident workgroup (10, 12, 14, 16) instead of letting the library steer it (measurement)")
    ap.add_argument("--collapse-weights", type=float, nargs=2, default=None, help="weights (area, triangle count) of the 4-wide collapse's objective (measurement; rtggx_debug_collapse_weights)")
    ap.add_argument("--tile-words", choices=["on", "off"], default="on", help="off: every tile of the visibility target counts as drawn, as in rounds 1-3 (measurement)")
    ap.add_argument("--tone-map", choices=["auto", "fused", "two"], default="auto", help="temporal pass + tone map as one kernel or two (measurement; auto: the library's choice, fused on small launches)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sustained-frames", type=int, default=1024, help="N = 1: frames of a second, longer window after the timed one, reported as `sustained` (0: none)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="host threads of the cpu_baseline leg (default: the job's CPU share, at most 16)")
    ap.add_argument("--cpu-frames", type=int, default=8, help="timed oracle frames on all threads (one more, untimed, first); the single-thread leg times 1")
    ap.add_argument("--prime-frames", type=int, default=256, help="frames rendered during SET-UP, before the warm-up steps: the GPU's compute clock ramps for "
                    "30-50 ms after any idle period (profiles/r02_b_clock_ramp.txt), and the renderer's adaptive state settles over its first frames; 0 = none")
    ap.add_argument("--no-balance", action="store_true", help="N > 1: equal strips instead of strips balanced by covered pixels")
    ap.add_argument("--stub", action="store_true", help="no GPU: a stand-in renderer over gloo, to exercise the launcher and the multi-rank protocol (tests)")
    return ap.parse_args(argv)


# =====================================================================================================================
# launcher: `python bench.py --gpus N` without a launcher's environment
# =====================================================================================================================
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args):
    """Start args.gpus children -- fresh interpreters, one rank each, LOCAL_RANK = its GPU -- and relay rank 0's output.
    This parent imports neither torch nor the HIP library: nothing here ever touches the GPU, and no process that has is
    re-executed."""
    port = _free_port()
    procs = []
    for rank in range(args.gpus):
        env = dict(os.environ)
        env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(args.gpus), "LOCAL_WORLD_SIZE": str(args.gpus),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL, stderr=None, text=True))
    out, _ = procs[0].communicate()
    codes = [p.wait() for p in procs]
    line = next((l for l in reversed(out.splitlines()) if l.startswith("{")), None)
    if line is not None:
        print(line, flush=True)
    else:
        sys.stdout.write(out)
    worst = max((abs(c) for c in codes), default=0)
    if worst or line is None:
        raise SystemExit("bench.py: ranks exited with %s%s" % (codes, "" if line else " and rank 0 printed no result line"))


# =====================================================================================================================
# stand-in renderer (--stub): the multi-rank protocol of this file without a GPU
# =====================================================================================================================
class StubRenderer:
    """A strip "renderer" on the CPU: each frame fills its rows of a history and a back-buffer array with a function of
    (row, frame), then runs the real exchange plan of raytracedggx_amd.strips over gloo.  What it checks is what the launcher
    and the rank protocol must get right: every rank's apron rows arrive, rank 0 assembles the frame."""

    def __init__(self, width, height, rank, world, dist):
        import numpy as np
        import torch
        from raytracedggx_amd import strips
        self.np, self.torch, self.strips, self.dist = np, torch, strips, dist
        self.W, self.H, self.rank, self.world = width, height, rank, world
        self.bounds = None
        self.b, self.e = strips.strip_rows(height, rank, world)
        if world > 1 and self.e - self.b < strips.HISTORY_APRON:          # as StripRenderer does
            raise ValueError("strips of %d rows are thinner than the %d-row history apron" % (self.e - self.b, strips.HISTORY_APRON))
        self.history = torch.zeros((height, width), dtype=torch.int64)
        self.back = torch.zeros((height, width), dtype=torch.int32)
        self.tokens = torch.zeros(2 * strips.capi.MAX_PEERS, dtype=torch.int32)      # the ordering tokens between ranks that exchange nothing else (strips.exchange_plan)
        self.frames = 0
        self.rays = 0
        self.plan = strips.exchange_plan(height, rank, world)

    def _truth(self, r0, r1, frame):
        return (self.torch.arange(r0, r1, dtype=self.torch.int64)[:, None] * 1000 + self.torch.arange(self.W, dtype=self.torch.int64)[None, :]) + frame

    def frame(self):
        f = self.frames
        self.history[self.b:self.e] = self._truth(self.b, self.e, f)
        self.back[self.b:self.e] = self._truth(self.b, self.e, f + 7).to(self.torch.int32)
        if self.world > 1:
            self.tokens[self.rank] = f + 1
            self.strips.run_exchange(self.dist, self.plan, {"history": self.history, "backbuffer": self.back, "token": self.tokens})
            for op, name, r0, r1, peer in self.plan:
                if op == "recv" and name == "token":
                    assert int(self.tokens[r0]) == f + 1, "rank %d: token of rank %d in frame %d" % (self.rank, peer, f)
            lo, hi = max(self.b - self.strips.HISTORY_APRON, 0), min(self.e + self.strips.HISTORY_APRON, self.H)
            assert self.torch.equal(self.history[lo:hi], self._truth(lo, hi, f)), "rank %d: history apron of frame %d" % (self.rank, f)
            if self.rank == 0:
                assert self.torch.equal(self.back, self._truth(0, self.H, f + 7).to(self.torch.int32)), "frame assembly on rank 0, frame %d" % f
        self.frames += 1
        self.rays += (self.e - self.b) * self.W // 4

    def rays_traced_since_reset(self):
        r, self.rays = self.rays, 0
        return r

    def close(self):
        pass


# =====================================================================================================================
# one rank
# =====================================================================================================================
def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)          # decided before torch is imported or any HIP call is made

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    W, H = args.width, args.height

    if args.stub:
        if world > 1:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        r = StubRenderer(W, H, rank, world, dist)
        device = "cpu"

        def barrier():
            if world > 1:
                dist.barrier()
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
        torch.cuda.set_device(local_rank)
        if world > 1:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        import assets
        from raytracedggx_amd.strips import StripRenderer
        r = StripRenderer(W, H, assets.path(args.mesh), assets.path("rnl_cross.dds"), rank=rank, world=world, device=local_rank,
                          dist=dist if world > 1 else None, balance=world > 1 and not args.no_balance,
                          extra_args=("-sharedmem",) + (("-metallic", args.metallic[0], args.metallic[1]) if args.metallic else ())
                          + (("-deform", args.deform) if args.deform else ()))
        ctx = r.context
        device = "cuda"
        peak_measured = None

        def barrier():
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            ctx.sync()

    # Set-up, part 2: priming.  Not a warm-up step of the contract and not timed: the same frames as below, rendered until the GPU's
    # clock governor has settled (it ramps the compute clock over 30-50 ms of sustained load after ANY idle period -- set-up leaves the
    # GPU idle for about a second -- which costs the first ~150 frames 5-13 %; measured in profiles/r02_b_clock_ramp.txt).  Reported
    # in the result line as config.setup_priming_frames.
    if not args.stub and args.trace_waves:
        ctx.trace_residency(args.trace_waves)
    if not args.stub and args.collapse_weights:
        ctx.collapse_weights(*args.collapse_weights); ctx.build_as()
    if not args.stub and args.tile_words == "off":
        ctx.tile_words(False)
    if not args.stub and args.tone_map != "auto":
        ctx.fuse_tone_map(args.tone_map == "fused")
    for _ in range(0 if args.stub else args.prime_frames):
        r.frame()
    if not args.stub:
        # the box's attainable HBM bandwidth, quoted beside the vendor peak: a float4 copy kernel, 2 x 1 GiB per launch, 128 launches
        # (>= 50 ms) AFTER the priming frames -- i.e. with the compute clock ramped like the timed region's (round 2 measured 8 launches
        # right after set-up, inside the clock ramp, and read 4.6 TB/s)
        if rank == 0:
            ctx.sync()
            peak_measured = ctx.copy_bandwidth(1 << 30, 32)
        for _ in range(16):       # (the copy evicted everything: a few frames to refill the caches before the warm-up steps)
            r.frame()
    for _ in range(args.warmup):
        r.frame()
    if not args.stub:
        ctx.enable_timing(3)        # a HIP event pair around the ray-trace kernel of every 8th frame, no host sync
    r.rays_traced_since_reset()     # zero the running ray total
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r.frame()
    barrier()
    dt = time.perf_counter() - t0
    # per-frame ray counts and kernel durations were recorded without host synchronisation; collect them now
    rays_total = r.rays_traced_since_reset()
    own_rays = rays_total
    overreach = 0 if args.stub else r.history_overreach()
    if world > 1:
        t = torch.tensor([dt, float(rays_total), float(overreach)], dtype=torch.float64, device=device)
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, rays_total, overreach = float(tmax[0].item()), float(tsum[1].item()), int(tmax[2].item())

    if rank == 0:
        ms_per_step = dt * 1e3 / args.steps
        value = rays_total / dt / 1e6
        out = {
            "metric": "Mrays/s + ms/frame, bunny 1920x1080 1spp+denoise, 1/2/4/8 GPUs",
            "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
        }
        if args.stub:
            out["config"] = {"workload": "STUB: no rendering -- launcher / rank-protocol rehearsal over gloo, %dx%d" % (W, H), "parallelism": "row strips x%d" % world}
            out["data"] = "none (stub)"
        else:
            out.update(report(r, args, W, H, world, ms_per_step, rays_total, own_rays, overreach, peak_measured, np))
            # One more figure beside the contract's (N = 1 only; after everything above has been read): 1024 more frames.  The pipeline has
            # two stable states; a short window after a barrier mostly reads the faster one, a long run spends most of its
            # time in the slower one (profiles/r04_k_states.txt).
            if world == 1 and args.sustained_frames > 0:
                barrier(); ts = time.perf_counter()
                for _ in range(args.sustained_frames):
                    r.frame()
                barrier()
                out["sustained"] = {"frames": args.sustained_frames, "ms_per_step": round((time.perf_counter() - ts) * 1e3 / args.sustained_frames, 4),
                                    "note": "after the timed region, same loop, bracketed the same way: what a long run reads (the pipeline's slower state included); `value` and `ms_per_step` are the contract's K steps"}
        print(json.dumps(out), flush=True)
    r.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def report(r, args, W, H, world, ms_per_step, rays_total, own_rays, overreach, peak_measured, np):
    import glob
    kernel_ms = r.ray_kernel_ms_since_reset()
    rows = r.strip_rows_with_apron()
    rays_per_launch = own_rays / max(args.steps, 1)     # rays of this rank's strip (apron rays of a strip are not counted)
    alg_bytes = r.trace_kernel_algorithmic_bytes(rays_per_launch)              # 40 B per ray (32 read + 8-byte key) + the tree once
    alg_bytes_r01 = r.trace_kernel_algorithmic_bytes(rays_per_launch, per_ray=72)   # round 1's yardstick (64-byte record), for continuity
    diffuse = bool(args.metallic) and min(args.metallic) < 1.0
    frame_bytes = r.frame_algorithmic_bytes(rows, metallic_lt_1=diffuse)                      # the passes this frame launches
    frame_bytes_survey = r.frame_algorithmic_bytes(rows, metallic_lt_1=diffuse, survey=True)  # 146 B/px, incl. the diffuse passes an all-metal frame skips
    k_ms = float(np.mean(kernel_ms)) if len(kernel_ms) else float("nan")
    # HBM bytes per launch from hardware counters: collected in separate rocprofv3 --pmc runs of this same workload
    # (tools/pmc.sh), committed under profiles/; null for other workloads
    traffic, traffic_source, valu, pmc_name, l1 = None, None, None, None, None
    tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if tfiles and (W, H, args.mesh, world, args.metallic, args.deform) == (1920, 1080, "bunny.obj", 1, None, 0.0):
        with open(tfiles[-1]) as f:
            pmc = json.load(f)
        pmc_name = os.path.basename(tfiles[-1])
        traffic = pmc["kernels"].get("rt::traceKernel", {}).get("traffic_bytes")
        valu = pmc.get("valu_instructions_per_frame")
        l1 = pmc.get("trace_kernel_l1")      # vector-L1 requests of the trace kernel per CU and cycle (tools/make_profiles.sh)
        traffic_source = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 read correction)" % pmc_name
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms == k_ms and k_ms > 0 else None
    # The same kernel with nothing beside it (after the timed region: 32 frames issued to ONE stream, rtggx_set_async_compute(0)): in the
    # timed region it shares the chip with two other stages at low stream priority, and its duration there is not its speed.
    residency = None if args.stub else r.context.trace_residency(args.trace_waves)
    k_alone = float("nan")
    if world == 1:
        r.context.set_async_compute(False)
        for _ in range(8):
            r.render()
        r.context.enable_timing(2)
        for _ in range(32):
            r.render()
        alone = r.ray_kernel_ms_since_reset()
        r.context.enable_timing(0)
        r.context.set_async_compute(True)
        k_alone = float(np.mean(alone)) if len(alone) else float("nan")
    frame_gbs = frame_bytes / (ms_per_step * 1e-3) / 1e9
    out = {
        "config": {"workload": "configs[1]: %s + rnl_cross env, %dx%d, 1spp GGX reflection + full denoise chain (refl H,V; diff H,V, shared-memory variant; temporal; tone map), "
                               "%s, dt=1/60%s" % (args.mesh, W, H, "metallic %g %g" % tuple(args.metallic) if args.metallic else "all-metal default materials",
                                                  ", model deforming every frame (amplitude %g, asynchronous BVH refit)" % args.deform if args.deform else ""),
                   "rays_per_frame": round(rays_total / args.steps, 1), "parallelism": "row strips x%d" % world, "strip_bounds": r.bounds,
                   "history_apron_rows": r.apron, "history_overreach_rows": overreach,
                   "setup_priming_frames": args.prime_frames,
                   "trace_workgroup_waves": None if residency is None else residency[0], "trace_share_of_period": None if residency is None else round(residency[1], 3)},
        "roofline": {"bound": "hbm", "kernel": "rt::traceKernel", "achieved": None if achieved is None else round(achieved, 2),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None if achieved is None else round(achieved / HBM_PEAK_GBS, 5),
                     "traffic": traffic, "traffic_source": traffic_source, "algorithmic_bytes_per_launch": int(alg_bytes),
                     "algorithmic_bytes_per_ray": 40, "algorithmic_bytes_r01": int(alg_bytes_r01), "l1_requests": l1, "kernel_ms": round(k_ms, 4),
                     "kernel_ms_alone": None if k_alone != k_alone else round(k_alone, 4),
                     # rays / trace-kernel time (the line's `value` is rays / whole-frame time)
                     "trace_kernel_mrays": None if not (k_ms == k_ms and k_ms > 0) else round(rays_per_launch / (k_ms * 1e-3) / 1e6, 1),
                     "trace_kernel_mrays_alone": None if k_alone != k_alone else round(rays_per_launch / (k_alone * 1e-3) / 1e6, 1),
                     "frac_alone": None if k_alone != k_alone else round(alg_bytes / (k_alone * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                     "peak_measured": None if peak_measured is None else round(peak_measured, 1),
                     "peak_measured_how": "float4 copy kernel, 2 x 1 GiB per launch, best of 2 / 4 / 8 / 16 workgroups per CU, plain and with non-temporal loads / stores, 32 launches each (>= 12 ms), read + written bytes (rtggx_copy_bandwidth), on this box after the set-up priming frames",
                     "note": "traversal is a dependent gather from an L2-resident tree, not a stream: the fraction is reported because the contract asks for it. "
                             "kernel_ms is the duration DURING the timed region, beside two other pipeline stages and at low stream priority (the frame is bound by "
                             "wave-slot time, so the launch is tuned for few wave-cycles, not for its own duration); kernel_ms_alone is the same launch with the chip to "
                             "itself. DESIGN.md 'Roofline accounting', profiles/r02_d_limiter.txt",
                     "frame": {"algorithmic_bytes": int(frame_bytes), "bytes_per_pixel": 178 if diffuse else 118, "achieved": round(frame_gbs, 2), "frac": round(frame_gbs / HBM_PEAK_GBS, 5),
                               "frac_of_measured_peak": None if not peak_measured else round(frame_gbs / peak_measured, 5),
                               "algorithmic_bytes_survey": int(frame_bytes_survey), "bytes_per_pixel_survey": 178 if diffuse else 146,
                               "frac_survey": round(frame_bytes_survey / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                               "note": "algorithmic_bytes counts the passes this frame LAUNCHES (all-metal: the two diffuse filter passes, 28 B/px of SURVEY 8(d)'s 146, are skipped); *_survey is SURVEY's sum"}},
        # one EXTRA, fully instrumented frame after the timed region: an event before and after every pass.  These are per-pass
        # latencies of a single frame with nothing overlapping it (queueing included) -- not the throughput figures above.
        "instrumented_frame_ms": dict(r.last_timings(), note="one extra frame with events around every pass, nothing overlapped: latencies, not throughput; 'frame' = first pass start to tone map end"),
    }
    if valu:      # share of the frame's VALU issue slots in use (wave-level instructions x cycles each over 1024 SIMDs at 2.4 GHz)
        issue_ms = valu * VALU_CYCLES_PER_WAVE_INSTRUCTION / 1024.0 / SHADER_CLOCK_HZ * 1e3
        out["roofline"]["frame"]["valu_issue"] = {"instructions": valu, "cycles_per_instruction": VALU_CYCLES_PER_WAVE_INSTRUCTION, "issue_ms": round(issue_ms, 4),
                                                   "frac_of_frame": round(issue_ms / ms_per_step, 4),
                                                   "source": "profiles/%s (rocprofv3 --pmc SQ_INSTS_VALU, summed over the kernels of one frame); cycles per instruction: MI355X_MICROARCH.md constants table, confirmed by tools/microbench/valu_issue.hip" % pmc_name}
    if world == 1 and not args.no_cpu_baseline and not args.deform:
        out["cpu_baseline"] = cpu_baseline(r, args, W, H)
    return out


def cpu_baseline(r, args, W, H):
    """The oracle (scalar C++ restatement, `port`) on the host cores: same inputs, same BVH arrays, same frames 0..n-1.
    Two legs: all hardware threads (row-interleaved pool), then one thread."""
    import assets
    from oracle import oracle as O
    from raytracedggx_amd import app, capi
    # How many host threads?  BASELINE.md 2 says "all hardware threads"; what a job on the GPU box may USE is its CPU share, not the host's
    # core count: the box gives one GPU's job 16 CPUs (os.cpu_count() reports the host's 256).  The share is read from the scheduler
    # affinity and the cgroup's CPU quota; the main leg runs on min(share, 16) threads (--cpu-threads overrides), and when the share is
    # larger than that a second leg runs on all of it.  Both are in the line, with what was found.
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, period = f.read().split()
            quota = None if q == "max" else max(1, int(int(q) / int(period)))
    except (OSError, ValueError):
        pass
    share = min(affinity, quota) if quota else affinity
    cores = args.cpu_threads if args.cpu_threads else min(share, 16)

    def leg(threads, frames):
        o = O.Oracle(W, H, threads=threads)
        v, i, _ = O.obj_import(assets.path(args.mesh))
        o.set_mesh(1, v, i)
        o.set_env_dds(assets.path("rnl_cross.dds"))
        if args.metallic:
            o.set_metallic(0, args.metallic[0]); o.set_metallic(1, args.metallic[1])
        for slot, (bn, bt) in enumerate(((capi.BUF_BVH_NODES0, capi.BUF_BVH_TRIS0), (capi.BUF_BVH_NODES1, capi.BUF_BVH_TRIS1))):
            o.set_bvh(slot, r.context.readback(bn), r.context.readback(bt), r.context.bvh_root(slot))
        o.transform_sh()
        fc = app.frame_constants(W, H, 1 + frames)
        rays, t_total, t_trace = 0, 0.0, 0.0
        for f in range(1 + frames):
            o.set_frame_constants(fc[f].tobytes()[:704] + o.get_frame_constants().tobytes()[704:])
            t0 = time.perf_counter()
            o.update_as(); o.render_visibility()
            t1 = time.perf_counter()
            n = o.ray_trace()
            t2 = time.perf_counter()
            o.denoise(); o.tone_map()
            t3 = time.perf_counter()
            if f > 0:   # frame 0 is the warm-up
                rays += n; t_total += t3 - t0; t_trace += t2 - t1
        o.close()
        return {"value": round(rays / t_total / 1e6, 4), "ms_per_frame": round(t_total * 1e3 / frames, 2), "trace_only_mrays": round(rays / t_trace / 1e6, 4)}

    multi = leg(cores, args.cpu_frames)
    every = leg(share, args.cpu_frames) if share > cores and not args.cpu_threads else None
    single = leg(1, 1)
    cpu = ""
    try:
        with open("/proc/cpuinfo") as f:
            cpu = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "")
    except OSError:
        pass
    return {"value": multi["value"], "unit": "Mrays/s", "cores": cores, "kind": "port", "ms_per_frame": multi["ms_per_frame"],
            "trace_only_mrays": multi["trace_only_mrays"], "cpu": cpu, "host_threads_available": os.cpu_count(), "cpu_share_of_this_job": share,
            "cpu_share_how": "min(scheduler affinity %d, cgroup cpu.max %s)" % (affinity, quota if quota else "unlimited"),
            "all_threads_of_the_share": None if every is None else dict(every, cores=share),
            "single_thread": dict(single, cores=1, sample="1 frame after 1 warm-up frame"),
            "sample": "%d frames (after 1 warm-up) of the same %dx%d workload, oracle on %d threads, same BVH arrays as the GPU; then 1 frame on 1 thread" % (args.cpu_frames, W, H, cores)}


if __name__ == "__main__":
    main()
