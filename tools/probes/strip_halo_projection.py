"""Strip overhead, the design question of VERDICT r01 item 6: a rank's strip needs +-18 rows of G-buffer and H-filter scratch beyond
its own rows for the vertical filters.  The build RECOMPUTES them (visibility, ray generation, traversal, shading and the H filters
cover own rows + 36; nothing but last frame's history crosses ranks, off the frame's critical path).  The north-star sketch
EXCHANGES them (own rows only, then 2 x 18 halo rows of normal 4 + roughness/metal 2 + depth 8 + H scratch 8 + raw reflection 4 B/px
from each neighbour, twice per frame, ON the critical path).  This probe times, one strip at a time on one GPU, the compute of
both: `recompute` as shipped, `receive` = RTGGX_GBUFFER_APRON=0 (own rows only; the halo rows are simply missing, so the images are
wrong near the edges -- it is a timing of the work that design leaves on a rank), plus the device-to-device copy of the halo bytes
as a lower bound of the exchange itself.  The slowest strip bounds the N-GPU frame rate from above.
   RTGGX_GBUFFER_APRON=0 python tools/probes/strip_halo_projection.py 1920 1080        (and once without the variable)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import assets
from raytracedggx_amd.strips import StripRenderer, balanced_bounds
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
mode = "receive (own rows only)" if os.environ.get("RTGGX_GBUFFER_APRON") == "0" else "recompute (own rows + 36)"
mesh, env = assets.path("bunny.obj"), assets.path("rnl_cross.dds")
print("%dx%d, %s" % (W, H, mode), flush=True)
for N in (1, 2, 4, 8):
    bounds = None
    if N > 1:      # the balanced boundaries bench.py uses
        p = StripRenderer(W, H, mesh, env, rank=0, world=N, transport=lambda *_: None, extra_args=("-sharedmem",), balance=True)
        bounds = p.bounds; p.close()
    times = []
    for r in range(N):
        s = StripRenderer(W, H, mesh, env, rank=r, world=N, transport=(lambda *_: None) if N > 1 else None, extra_args=("-sharedmem",), balance=bounds if N > 1 else False)
        for _ in range(200): s.render()
        s.context.sync()
        t0 = time.perf_counter()
        for _ in range(200): s.render()
        s.context.sync()
        times.append((time.perf_counter() - t0) / 200 * 1e3)
        s.close()
    print("N=%d: slowest strip %.4f ms  (strips: %s)  bounds %s" % (N, max(times), " ".join("%.3f" % t for t in times), bounds), flush=True)
# the exchange the `receive` design adds per neighbour and frame: 18 rows x W x 26 B, as two messages (G-buffer after ray generation, H scratch after
# the H pass); on one GPU only the copy itself can be timed
nbytes = 18 * W * 26
a, b = torch.empty(nbytes, dtype=torch.uint8, device="cuda"), torch.empty(nbytes, dtype=torch.uint8, device="cuda")
for _ in range(20): b.copy_(a)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): b.copy_(a)
torch.cuda.synchronize()
print("halo bytes per neighbour and frame: %d; device copy of that: %.1f us (the real exchange adds RCCL's launch and xGMI latency twice per frame)" % (nbytes, (time.perf_counter() - t0) / 200 * 1e6))
