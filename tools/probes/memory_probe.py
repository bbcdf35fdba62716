"""Device memory a context takes (hipMemGetInfo before the context exists and after 16 frames), by frame size and materials.
   python tools/probes/memory_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import assets
from raytracedggx_amd import app
torch.cuda.init()
for mesh, W, H, extra in (("bunny.obj", 1920, 1080, ()), ("bunny.obj", 1920, 1080, ("-metallic", 1.0, 0.5)), ("dragon.obj", 1920, 1080, ()), ("bunny.obj", 3840, 2160, ()), ("bunny.obj", 1920, 171, ())):
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    a = app.RayTracedGGX(["-mesh", assets.path(mesh), "-env", assets.path("rnl_cross.dds"), "-width", W, "-height", H, "-sharedmem"] + list(extra))
    for _ in range(16):
        a.OnUpdate(); a.OnRender()
    a.context.sync()
    free1, _ = torch.cuda.mem_get_info()
    print("%-10s %4dx%-4d %-20s %7.1f MB" % (mesh, W, H, " ".join(str(x) for x in extra) or "all-metal", (free0 - free1) / 1e6), flush=True)
    a.OnDestroy()
