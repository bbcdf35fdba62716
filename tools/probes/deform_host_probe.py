"""Where the host spends a deforming frame, per 50-frame window: wall clock around OnUpdate (the vertex staging of rtggx_refit_as) and each
C-ABI call of the frame, beside the window's frame time -- to see what differs between the pipeline's two states
(profiles/r03_i_deform_states.txt).   python tools/probes/deform_host_probe.py [mesh] [W H] [frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import assets
from raytracedggx_amd import app
mesh = sys.argv[1] if len(sys.argv) > 1 else "bunny.obj"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 1500
a = app.RayTracedGGX(["-mesh", assets.path(mesh), "-env", assets.path("rnl_cross.dds"), "-width", W, "-height", H, "-sharedmem", "-deform", 0.3])
c = a.context
for _ in range(200): a.OnUpdate(); a.OnRender()
c.sync()
calls = (("update_as", c.update_as), ("render_visibility", c.render_visibility), ("ray_trace", c.ray_trace), ("denoise", lambda: c.denoise(True)), ("tone_map", c.tone_map))
print("window  ms/frame | us per frame on the host: OnUpdate " + " ".join(n for n, _ in calls))
for w in range(frames // 50):
    acc = [0.0] * (1 + len(calls))
    t0 = time.perf_counter()
    for _ in range(50):
        t = time.perf_counter(); a.OnUpdate(); acc[0] += time.perf_counter() - t
        for i, (_, fn) in enumerate(calls):
            t = time.perf_counter(); fn(); acc[1 + i] += time.perf_counter() - t
    dt = time.perf_counter() - t0
    fw_us, fw_n = c.fence_wait()
    c.sync()
    print("%4d   %.4f   | " % (w, dt / 50 * 1e3) + "  ".join("%6.1f" % (v / 50 * 1e6) for v in acc) + "   host total %.1f, of it waiting at the fence %.1f (%d of 50 frames)" % (sum(acc) / 50 * 1e6, fw_us / 50, fw_n), flush=True)
a.OnDestroy()
