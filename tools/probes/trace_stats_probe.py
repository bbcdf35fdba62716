"""Manual probe: per-frame counters of a -DRT_TRACE_STATS build of librtggx
(make -C raytracedggx_amd EXTRA=-DRT_TRACE_STATS after removing _build/trace.o)."""
import sys
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0]); sys.path.insert(0, __file__.rsplit("/tools/", 1)[0] + "/tests")
import assets
from raytracedggx_amd import app
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
a = app.RayTracedGGX(["-mesh", assets.path("bunny.obj"), "-env", assets.path("rnl_cross.dds"), "-width", W, "-height", H])
a.context.enable_timing(1)
for f in range(int(sys.argv[3]) if len(sys.argv) > 3 else 3):
    a.OnUpdate(); a.OnRender(); a.context.sync()
    allc = a.context.debug_counters(768, reset=True)
    raw = allc[:256].astype(float).reshape(16, 16)   # 16 copies of the counters
    c = raw.sum(axis=0); c[[4, 6, 7]] = raw.max(axis=0)[[4, 6, 7]]
    rays = a.context.ray_count()
    waves = max(c[3], 1)
    print("frame %d rays %d  node steps/ray %.1f  leaf steps/ray %.2f  wave iterations %d (%.1f per wave, lane utilisation %.2f)  waves %d  deepest stack %d  steals %d  leaf phases %d  kernel %.3f ms" % (
        f, rays, c[0] / rays, c[1] / rays, c[2], c[2] / waves, (c[0] + c[1]) / max(c[2] * 64.0, 1), c[3], c[4], c[8], c[9], a.context.timings()["ray_trace_kernel"]))
    print("   wave lifetime: mean %.0f cycles, max %.0f cycles; most iterations in one wave %d; cycles per iteration %.0f" % (
        c[5] * 1024 / waves, c[6] * 16, c[7], c[5] * 1024 / max(c[2], 1)))
    longest = max((int(raw[k, 11]), int(raw[k, 10])) for k in range(16))
    print("   the wave with the most iterations (%d) did %d lane steps: lane utilisation %.2f" % (longest[0], longest[1], longest[1] / max(longest[0] * 64.0, 1)))
    print("   wave starts per 8 us:", " ".join(str(int(v)) for v in allc[288:320]))
    print("   wave ends   per 8 us:", " ".join(str(int(v)) for v in allc[256:288]))
    for inst in range(2):
        lv = allc[512 + 32 * inst:544 + 32 * inst].astype(float)
        if lv.sum() > 0:
            print("   instance %d node steps by 4-wide level (share, cumulative):" % inst, " ".join("%d:%.1f%%/%.0f%%" % (k, 100 * lv[k] / c[0], 100 * lv[:k + 1].sum() / c[0]) for k in range(32) if lv[k] > 0))
import numpy as np
for inst, bid in ((0, app.capi.BUF_BVH4_NODES0), (1, app.capi.BUF_BVH4_NODES1)):
    n4 = a.context.readback(bid).reshape(-1, 32)
    used = n4.any(axis=1)
    pop = np.bincount(n4[used, 28].astype(np.int64), minlength=1)
    print("instance %d: %d 4-wide nodes; per level:" % (inst, used.sum()), " ".join(str(int(v)) for v in pop), "; cumulative:", " ".join(str(int(v)) for v in np.cumsum(pop)))
