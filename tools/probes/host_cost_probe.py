"""Host time per frame, per entry point of the C ABI (wall clock around each ctypes call, 3000 free-running frames): what a thin strip
of an 8-GPU frame is bound by (profiles/r02_c_ab_pipeline.txt: 256x144 55 us = render_visibility 16 + ray_trace 21 + denoise 11 +
tone_map 4, about 4 us per kernel launch).   python tools/probes/host_cost_probe.py [W H]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import assets
from raytracedggx_amd import app
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 144)
a = app.RayTracedGGX(["-mesh", assets.path("bunny.obj"), "-env", assets.path("rnl_cross.dds"), "-width", W, "-height", H, "-sharedmem"])
c = a.context
for _ in range(300): a.OnUpdate(); a.OnRender()
c.sync()
N = 3000
acc = {"OnUpdate": 0.0, "update_as": 0.0, "render_visibility": 0.0, "ray_trace": 0.0, "denoise": 0.0, "tone_map": 0.0}
t_all = time.perf_counter()
for _ in range(N):
    t = time.perf_counter(); a.OnUpdate(); acc["OnUpdate"] += time.perf_counter() - t
    for name, fn in (("update_as", c.update_as), ("render_visibility", c.render_visibility), ("ray_trace", c.ray_trace), ("denoise", lambda: c.denoise(True)), ("tone_map", c.tone_map)):
        t = time.perf_counter(); fn(); acc[name] += time.perf_counter() - t
total = time.perf_counter() - t_all
c.sync()
print("%dx%d: %.1f us per frame on the host; " % (W, H, total / N * 1e6) + ", ".join("%s %.1f" % (k, v / N * 1e6) for k, v in acc.items()))
a.OnDestroy()
