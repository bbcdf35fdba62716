"""Frame time and shader clock over a long free-running run (round 4): windows of 64 frames of the 1080p bunny, each timed on the host
(sync at both ends) and followed by a reading of the shader clock (rtggx_debug_shader_clock: s_memtime ticks per 100 MHz wall-clock tick
in a one-wave kernel); with a pause in the middle.  Are the pipeline's "two stable states"  two clock states?
python tools/probes/clock_states.py [windows] [pause_ms]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import assets
from raytracedggx_amd import app
windows = int(sys.argv[1]) if len(sys.argv) > 1 else 60
pause = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
a = app.RayTracedGGX(["-mesh", assets.path("bunny.obj"), "-env", assets.path("rnl_cross.dds"), "-width", 1920, "-height", 1080, "-sharedmem"])
c = a.context
for _ in range(8): a.OnUpdate(); a.OnRender()
c.sync()
t_start = time.perf_counter()
def run(n):
    for w in range(n):
        t0 = time.perf_counter()
        for _ in range(64): a.OnUpdate(); a.OnRender()
        c.sync()
        ms = (time.perf_counter() - t0) / 64 * 1e3
        mhz = c.shader_clock_mhz()
        print("t %7.1f ms  window %3d  %.4f ms/frame  shader clock %6.0f MHz" % ((time.perf_counter() - t_start) * 1e3, w, ms, mhz), flush=True)
run(windows)
print("-- idle for %.0f ms --" % pause); time.sleep(pause * 1e-3)
run(windows // 2)
a.OnDestroy()
