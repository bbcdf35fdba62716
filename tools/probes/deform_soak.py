"""Frame time of a deforming mesh over a long run: N frames of `-deform amp` (new vertices + asynchronous refit every frame; the library
rebuilds the tree beside the frames when its cost has drifted), the mean of every 50 frames, refits / rebuilds / cost ratio along the way.
   python tools/probes/deform_soak.py [mesh] [W H] [frames] [amp] [rebuild_ratio] [steps_per_frame]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import assets
from raytracedggx_amd import app
mesh = sys.argv[1] if len(sys.argv) > 1 else "dragon.obj"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
amp = float(sys.argv[5]) if len(sys.argv) > 5 else 0.3
a = app.RayTracedGGX(["-mesh", assets.path(mesh), "-env", assets.path("rnl_cross.dds"), "-width", W, "-height", H, "-sharedmem", "-deform", amp])
c = a.context
if len(sys.argv) > 7:
    c.set_refit_policy(float(sys.argv[6]), int(sys.argv[7]))
for _ in range(200):
    a.OnUpdate(); a.OnRender()
c.sync()
chunk, means, worst = 50, [], []
for k in range(frames // chunk):
    t0 = time.perf_counter()
    tp, per = t0, []
    for _ in range(chunk):
        a.OnUpdate(); a.OnRender()
        t = time.perf_counter(); per.append((t - tp) * 1e3); tp = t
    c.sync()
    means.append((time.perf_counter() - t0) / chunk * 1e3)
    worst.append(max(per))
st = c.refit_stats(1)
m = np.array(means)
print("%s %dx%d deform %g, policy %s: %d frames, ms per frame over %d-frame windows: mean %.4f  min %.4f  max %.4f  (-%.1f %% / +%.1f %% of the mean); refits %d, rebuilds %d, cost ratio now %.2f" % (
    mesh, W, H, amp, sys.argv[6:8] or "default", frames, chunk, m.mean(), m.min(), m.max(), 100 * (1 - m.min() / m.mean()), 100 * (m.max() / m.mean() - 1), st["refits"], st["rebuilds"], st["cost_ratio"]))
print("   windows: " + " ".join("%.3f" % x for x in means))
print("   longest single frame on the host per window (ms): " + " ".join("%.2f" % x for x in worst))
a.OnDestroy()
