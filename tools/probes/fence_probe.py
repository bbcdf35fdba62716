"""Host time per entry point and the frames-in-flight fence on a full-size frame (1080p bunny, 1024 free-running frames): is the host or the GPU the
limit?  (Round 4: 0.184 ms per frame; render_visibility 144 us of which 131 waiting at the fence in 1019 of 1024 frames, everything else 38 us: the
host is parked at the fence every frame -- a frame enters the pipeline when the one four frames ahead of it leaves.)   python tools/probes/fence_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import assets
from raytracedggx_amd import app
a = app.RayTracedGGX(["-mesh", assets.path("bunny.obj"), "-env", assets.path("rnl_cross.dds"), "-width", 1920, "-height", 1080, "-sharedmem"])
c = a.context
for _ in range(300): a.OnUpdate(); a.OnRender()
c.sync(); c.fence_wait(True)
N = 1024
acc = {"OnUpdate": 0.0, "update_as": 0.0, "render_visibility": 0.0, "ray_trace": 0.0, "denoise": 0.0, "tone_map": 0.0}
t0 = time.perf_counter()
for _ in range(N):
    t = time.perf_counter(); a.OnUpdate(); acc["OnUpdate"] += time.perf_counter() - t
    for name, fn in (("update_as", c.update_as), ("render_visibility", c.render_visibility), ("ray_trace", c.ray_trace), ("denoise", lambda: c.denoise(True)), ("tone_map", c.tone_map)):
        t = time.perf_counter(); fn(); acc[name] += time.perf_counter() - t
c.sync()
tot = time.perf_counter() - t0
us, n = c.fence_wait(True)
print("%.4f ms/frame; host per call (us): %s; fence: %d of %d frames waited, %.1f us per frame" % (tot / N * 1e3, ", ".join("%s %.1f" % (k, v / N * 1e6) for k, v in acc.items()), n, N, us / N))
a.OnDestroy()
