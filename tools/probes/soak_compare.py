"""Long free-running runs against the same frames synchronised one by one (the parity tests do 30-48 frames; this does thousands): every
target bit-identical at the end, checkpoints on the way.  Material changes on a schedule, so that the diffuse image's carry-over changes
hands (capi.hip rtggx_ray_trace) many times.   python tools/probes/soak_compare.py [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import assets
from raytracedggx_amd import app, capi
FRAMES = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
BUFS = (capi.BUF_VISIBILITY, capi.BUF_DEPTH, capi.BUF_NORMAL, capi.BUF_ROUGH_METAL, capi.BUF_VELOCITY, capi.BUF_RT_REFL, capi.BUF_RT_DIFF, capi.BUF_FLT_DFF, capi.BUF_TSS0, capi.BUF_TSS1, capi.BUF_BACKBUFFER)
bad = 0
for size, extra, every in (((320, 180), [], 7), ((640, 360), ["-metallic", 0.25, 0.5], 0), ((1920, 171), [], 11), ((1280, 720), [], 13), ((640, 360), ["-deform", 0.3], 0)):
    args = ["-mesh", assets.path("bunny.obj"), "-env", assets.path("rnl_cross.dds"), "-width", size[0], "-height", size[1], "-sharedmem", "-dt", 0.02] + extra
    a, b = app.RayTracedGGX(args), app.RayTracedGGX(args)
    try:
        for f in range(FRAMES):
            if every and f % every == 0:
                m = 1.0 if (f // every) % 3 else 0.5
                a.context.set_metallic(0, m); b.context.set_metallic(0, m)
            a.OnUpdate(); a.OnRender(); a.context.sync()
            b.OnUpdate(); b.OnRender()
            if (f + 1) % 500 == 0 or f + 1 == FRAMES:
                b.context.sync()
                diff = [bid for bid in BUFS if not np.array_equal(a.context.readback(bid), b.context.readback(bid))]
                print("%dx%d %s frame %d: %s" % (size[0], size[1], " ".join(str(x) for x in extra), f + 1, "identical" if not diff else "DIFFERENT buffers %s" % diff), flush=True)
                bad += bool(diff)
    finally:
        a.OnDestroy(); b.OnDestroy()
print("soak: %s" % ("ok" if not bad else "%d checkpoints differ" % bad))
sys.exit(1 if bad else 0)
