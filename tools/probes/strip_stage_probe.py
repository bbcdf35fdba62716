"""Per strip of an N-rank frame (balanced boundaries, one strip at a time on one GPU): frame time, rays, the trace kernel's duration
(HIP events on the dispatch) and the traversal's workgroup size / share of the period.  The numbers of profiles/r02_c_ab_pipeline.txt
("per strip of N = 2 / 4").   python tools/probes/strip_stage_probe.py [N]      STRIP_ONLY=r: that strip alone (for a kernel trace)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import assets
from raytracedggx_amd.strips import StripRenderer
W, H, N = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 2
mesh, env = assets.path("bunny.obj"), assets.path("rnl_cross.dds")
p = StripRenderer(W, H, mesh, env, rank=0, world=N, transport=lambda *_: None, extra_args=("-sharedmem",), balance=True)
bounds = p.bounds; p.close()
ONLY = os.environ.get("STRIP_ONLY")
for r in range(N):
    if ONLY is not None and r != int(ONLY): continue
    s = StripRenderer(W, H, mesh, env, rank=r, world=N, transport=lambda *_: None, extra_args=("-sharedmem",), balance=bounds)
    for _ in range(300): s.render()
    s.context.sync()
    s.context.enable_timing(3)
    s.rays_traced_since_reset()
    t0 = time.perf_counter()
    for _ in range(300): s.render()
    s.context.sync()
    ms = (time.perf_counter() - t0) / 300 * 1e3
    rays = s.rays_traced_since_reset() / 300
    k = s.ray_kernel_ms_since_reset()
    print("N=%d strip %d rows [%d,%d): %.4f ms/frame, %.0f rays, trace kernel %.4f ms, residency %s" % (N, r, s.b, s.e, ms, rays, sum(k) / max(len(k), 1), s.context.trace_residency()), flush=True)
    s.close()
