#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X.

  metric   : Mrays/s (+ ms/frame in `ms_per_step`), bunny + rnl env, 1920x1080, 1spp + full denoise chain
  step     : one frame = RayTracedGGX::OnUpdate + OnRender (visibility -> ray trace -> 4 spatial passes -> temporal -> tone map)
  N GPUs   : the SAME 1920x1080 frame sharded by row strips, one process per GPU (torchrun), history-apron
             exchange over RCCL + frame gather on rank 0 ("scaling": "strong")
  value    : non-degenerate rays traced by all ranks in the K timed frames / max-over-ranks wall time
  roofline : the dominant kernel (rt::traceKernel: BVH traversal of the binned rays), algorithmic bytes per launch
             (DESIGN.md "Roofline accounting") / its average duration from HIP events recorded on the launching stream
             (stream B); `traffic` = HBM-side bytes per launch from the newest profiles/*_pmc_traffic.json;
             `roofline.frame` = SURVEY 8(d)'s whole-frame bytes / frame time
  cpu_baseline : the scalar C++ oracle re-tracing the SAME BVH arrays on the host cores, bounded sample (rank 0, N=1 only)

Usage: python bench.py [--gpus N] [--steps K] [--warmup W] [--width 1920 --height 1080 --mesh bunny.obj]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--mesh", default="bunny.obj")
    ap.add_argument("--metallic", type=float, nargs=2, default=None, help="metallic of ground and model (default: the sample's 1 1 = no diffuse rays)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=2)
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import assets
    from raytracedggx_amd import capi
    from raytracedggx_amd.strips import StripRenderer

    W, H = args.width, args.height
    r = StripRenderer(W, H, assets.path(args.mesh), assets.path("rnl_cross.dds"), rank=rank, world=world, device=local_rank,
                      dist=dist if world > 1 else None, balance=world > 1,
                      extra_args=("-sharedmem",) + (("-metallic", args.metallic[0], args.metallic[1]) if args.metallic else ()))
    ctx = r.context

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.sync()

    for _ in range(args.warmup):
        r.frame()
    ctx.enable_timing(3)            # a HIP event pair around the ray-trace kernel of every 8th frame, no host sync
    r.rays_traced_since_reset()     # zero the device-side running ray total
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r.frame()
    barrier()
    dt = time.perf_counter() - t0
    # per-frame ray counts and kernel durations were recorded without host synchronisation; collect them now
    rays_total = r.rays_traced_since_reset()
    kernel_ms = r.ray_kernel_ms_since_reset()
    own_rays = rays_total

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        rt = torch.tensor([rays_total], dtype=torch.float64, device="cuda")
        dist.all_reduce(rt, op=dist.ReduceOp.SUM)
        rays_total = float(rt.item())

    if rank == 0:
        ms_per_step = dt * 1e3 / args.steps
        value = rays_total / dt / 1e6
        rows = r.strip_rows_with_apron()
        rays_per_launch = own_rays / max(args.steps, 1)     # rays of this rank's strip (apron rays of a strip are not counted)
        alg_bytes = r.trace_kernel_algorithmic_bytes(rays_per_launch)
        frame_bytes = r.frame_algorithmic_bytes(rows, metallic_lt_1=bool(args.metallic) and min(args.metallic) < 1.0)
        k_ms = float(np.mean(kernel_ms)) if len(kernel_ms) else float("nan")
        # HBM bytes per launch from hardware counters: collected in separate rocprofv3 --pmc runs of this same workload
        # (tools/pmc.sh), committed under profiles/; null for other workloads
        traffic, traffic_source, valu = None, None, None
        import glob
        tfiles = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_pmc_traffic.json")))
        if tfiles and (W, H, args.mesh, world) == (1920, 1080, "bunny.obj", 1):
            with open(tfiles[-1]) as f:
                pmc = json.load(f)
            traffic = pmc["kernels"].get("rt::traceKernel", {}).get("traffic_bytes")
            valu = pmc.get("valu_instructions_per_frame")
            traffic_source = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 read correction)" % os.path.basename(tfiles[-1])
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms == k_ms and k_ms > 0 else None
        out = {
            "metric": "Mrays/s + ms/frame, bunny 1920x1080 1spp+denoise, 1/2/4/8 GPUs",
            "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: %s + rnl_cross env, %dx%d, 1spp GGX reflection + full denoise chain (refl H,V; diff H,V, shared-memory variant; temporal; tone map), "
                                   "%s, dt=1/60" % (args.mesh, W, H, "metallic %g %g" % tuple(args.metallic) if args.metallic else "all-metal default materials"),
                       "rays_per_frame": round(rays_total / args.steps, 1), "parallelism": "row strips x%d" % world, "strip_bounds": r.bounds},
            "roofline": {"bound": "hbm", "kernel": "rt::traceKernel", "achieved": None if achieved is None else round(achieved, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None if achieved is None else round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": traffic, "traffic_source": traffic_source, "algorithmic_bytes_per_launch": int(alg_bytes), "kernel_ms": round(k_ms, 4),
                         "note": "traversal is not bandwidth-shaped: with both streams busy the frame is bound by VALU issue (~80 % of the issue slots of 1024 SIMDs, profiles/*_pmc_report.txt; DESIGN.md 'Roofline')",
                         "frame": {"algorithmic_bytes": int(frame_bytes), "achieved": round(frame_bytes / (ms_per_step * 1e-3) / 1e9, 2),
                                   "frac": round(frame_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}},
            "passes_ms": r.last_timings(),
        }
        if valu:      # what actually bounds the frame: VALU issue (wave-level instructions x 4 cycles over 1024 SIMDs at 2.4 GHz)
            issue_ms = valu * 4.0 / 1024.0 / 2.4e9 * 1e3
            out["roofline"]["frame"]["valu_issue"] = {"instructions": valu, "issue_ms": round(issue_ms, 4), "frac_of_frame": round(issue_ms / ms_per_step, 4),
                                                       "source": "profiles/%s (rocprofv3 --pmc SQ_INSTS_VALU, summed over the kernels of one frame)" % os.path.basename(tfiles[-1])}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(r, args, W, H)
        print(json.dumps(out), flush=True)
    r.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(r, args, W, H):
    """The oracle (scalar C++ restatement, `port`) on the host cores: same inputs, same BVH arrays, same frames 0..n-1."""
    import assets
    from oracle import oracle as O
    from raytracedggx_amd import app, capi
    cores = min(os.cpu_count() or 1, 16)
    o = O.Oracle(W, H, threads=cores)
    v, i, _ = O.obj_import(assets.path(args.mesh))
    o.set_mesh(1, v, i)
    o.set_env_dds(assets.path("rnl_cross.dds"))
    if args.metallic:
        o.set_metallic(0, args.metallic[0]); o.set_metallic(1, args.metallic[1])
    for slot, (bn, bt) in enumerate(((capi.BUF_BVH_NODES0, capi.BUF_BVH_TRIS0), (capi.BUF_BVH_NODES1, capi.BUF_BVH_TRIS1))):
        o.set_bvh(slot, r.context.readback(bn), r.context.readback(bt), r.context.bvh_root(slot))
    o.transform_sh()
    fc = app.frame_constants(W, H, 1 + args.cpu_frames)
    rays, t_total, t_trace = 0, 0.0, 0.0
    for f in range(1 + args.cpu_frames):
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
    return {"value": round(rays / t_total / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "ms_per_frame": round(t_total * 1e3 / args.cpu_frames, 2), "trace_only_mrays": round(rays / t_trace / 1e6, 4),
            "sample": "%d frames (after 1 warm-up) of the same %dx%d workload, oracle on %d threads, same BVH arrays as the GPU" % (args.cpu_frames, W, H, cores)}


if __name__ == "__main__":
    main()
