"""Manual probe: what the two weights of the 4-wide collapse's objective (rtggx_debug_collapse_weights: surface area / triangle count)
do to the tree and to the traversal -- nodes, entries per node, levels, node and leaf steps per ray (a -DRT_TRACE_STATS build of
librtggx: rm raytracedggx_amd/_build/trace.o && make -C raytracedggx_amd EXTRA=-DRT_TRACE_STATS; zeros otherwise), and the trace
kernel's time with nothing beside it.   python tools/probes/collapse_probe.py [mesh] [W H] [-metallic m0 m1]"""
import sys
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0]); sys.path.insert(0, __file__.rsplit("/tools/", 1)[0] + "/tests")
import numpy as np
import assets
from raytracedggx_amd import app, capi
mesh = sys.argv[1] if len(sys.argv) > 1 else "bunny.obj"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
extra = sys.argv[4:]
a = app.RayTracedGGX(["-mesh", assets.path(mesh), "-env", assets.path("rnl_cross.dds"), "-width", W, "-height", H, "-sharedmem"] + extra)
c = a.context
for wa, wt in ((1.0, 0.0), (1.0, 0.1), (1.0, 0.25), (1.0, 0.5), (1.0, 1.0), (1.0, 2.0), (1.0, 4.0), (0.25, 1.0), (0.0, 1.0)):
    c.collapse_weights(wa, wt)
    c.build_as()
    c.enable_timing(1)
    steps = []
    for f in range(4):
        a.OnUpdate(); a.OnRender(); c.sync()
        raw = c.debug_counters(768, reset=True)[:256].astype(float).reshape(16, 16).sum(axis=0)
        rays = c.ray_count()
        steps.append((raw[0] / rays, raw[1] / rays, c.timings()["ray_trace_kernel"]))
    c.enable_timing(0)
    n4 = c.readback(capi.BUF_BVH4_NODES1).reshape(-1, 32)
    used = n4.any(axis=1)
    entries = int((n4[used, 24:28].view(np.int32) != 0x7FFFFFFF).sum())
    s = np.array(steps[1:])
    print("%s %dx%d  area %.2f tris %.2f: %6d nodes, %.2f entries per node, %2d levels; node steps per ray %.2f, leaf steps %.2f; trace kernel alone %.4f ms" % (
        mesh, W, H, wa, wt, used.sum(), entries / used.sum(), int(n4[used, 28].max()) + 1, s[:, 0].mean(), s[:, 1].mean(), s[:, 2].mean()), flush=True)
a.OnDestroy()
