"""Does the pipeline's faster state come from power headroom (round 4, profiles/r04_k_states.txt)?  2048 free-running frames of the 1080p bunny,
with a pause of `gap_ms` after every `every` frames (0: none); RTGGX_TRACE_LOG=1 prints the period per 16-frame sample to stderr.
python tools/probes/idle_gaps.py every gap_ms"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import assets
from raytracedggx_amd import app
every, gap = int(sys.argv[1]), float(sys.argv[2])
a = app.RayTracedGGX(["-mesh", assets.path("bunny.obj"), "-env", assets.path("rnl_cross.dds"), "-width", 1920, "-height", 1080, "-sharedmem"])
c = a.context
for _ in range(256): a.OnUpdate(); a.OnRender()
c.sync()
busy = 0.0; t0 = time.perf_counter()
for f in range(2048):
    a.OnUpdate(); a.OnRender()
    if every and f % every == every - 1:
        c.sync(); busy += time.perf_counter() - t0; time.sleep(gap * 1e-3); t0 = time.perf_counter()
c.sync(); busy += time.perf_counter() - t0
print("every %d frames a pause of %.1f ms: %.4f ms per frame while busy" % (every, gap, busy / 2048 * 1e3))
a.OnDestroy()
