"""Where does the temporal result of a full-size frame differ from the oracle's?  Renders `frames` frames of a mesh at W x H through
tests' Pair (HIP path and oracle side by side) and prints, per floating-point target, the relative L2, and for the temporal result the
pixels that carry the difference.   python tools/probes/parity_probe.py [mesh] [W H] [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_parity as T
from oracle import oracle as O
mesh = sys.argv[1] if len(sys.argv) > 1 else "dragon.obj"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 2
p = T.Pair(W, H, mesh=mesh, shared_mem=True)
capi = p.capi
for f in range(frames):
    p.frame()
    par = p.ctx.frame_parity()
    for name, gid, oid in (("FilteredOut1", capi.BUF_FLT_DFF, O.BUF_FLT_DFF), ("TemporalSSOut", capi.BUF_TSS0 + par, O.BUF_TSS0 + par)):
        g, r = O.unpack_rgba16f(p.ctx.readback(gid)).astype(np.float64), O.unpack_rgba16f(p.o.buffer(oid)).astype(np.float64)
        fin = np.isfinite(r) & np.isfinite(g)
        d = np.where(fin, g - r, 0.0)
        num, den = (d ** 2).sum(), (np.where(fin, r, 0.0) ** 2).sum()
        print("frame %d %-14s rel L2 %.3e   non-finite: gpu %d oracle %d" % (f, name, np.sqrt(num / den), (~np.isfinite(g)).sum(), (~np.isfinite(r)).sum()))
        if name == "TemporalSSOut":
            e = (d[..., :3] ** 2).sum(axis=-1)
            order = np.argsort(e.ravel())[::-1]
            cum = np.cumsum(e.ravel()[order]) / max(e.sum(), 1e-300)
            print("   pixels carrying 50%% / 90%% / 99%% of the squared error: %d / %d / %d of %d;  per channel share of the error: %s;  alpha error %.3e" %
                  (np.searchsorted(cum, 0.5) + 1, np.searchsorted(cum, 0.9) + 1, np.searchsorted(cum, 0.99) + 1, e.size,
                   ["%.2f" % ((d[..., k] ** 2).sum() / max(num, 1e-300)) for k in range(4)], np.abs(d[..., 3]).max()))
            vis = p.ctx.readback(capi.BUF_VISIBILITY)
            for k in order[:12]:
                y, x = divmod(int(k), W)
                print("   (%4d,%4d) vis %08x  gpu %s  oracle %s" % (x, y, vis[y, x], np.array2string(g[y, x], precision=5), np.array2string(r[y, x], precision=5)))
            # without the brightest 0.1 % of the oracle's pixels
            lum = r[..., :3].sum(axis=-1); cut = np.quantile(lum[np.isfinite(lum)], 0.999)
            keep = fin.all(axis=-1) & (lum <= cut)
            print("   rel L2 without the brightest 0.1 %% of pixels: %.3e" % np.sqrt((d[keep] ** 2).sum() / (r[keep] ** 2).sum()))
p.close()
