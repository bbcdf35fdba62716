import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import test_gpu_parity as T
from oracle import oracle as O
p = T.Pair(16, 16, metallic=(0.5, 0.25), shared_mem=True)
p.frame()
capi = p.capi
g = O.unpack_rgba16f(p.ctx.readback(capi.BUF_FLT_RFL)); r = O.unpack_rgba16f(p.o.buffer(O.BUF_FLT_RFL))
bad = np.argwhere(~np.isfinite(r).all(axis=2))
print("oracle non-finite pixels:", len(bad), bad[:5])
for y, x in bad[:3]:
    print("pix", y, x, "gpu", g[y, x], "oracle", r[y, x])
    n = p.ctx.readback(capi.BUF_NORMAL)[y, x]; print("  normal word %08x" % n, "rm %04x" % p.ctx.readback(capi.BUF_ROUGH_METAL)[y, x], "depth", p.ctx.readback(capi.BUF_DEPTH)[y, x], "refl %08x" % p.ctx.readback(capi.BUF_RT_REFL)[y, x])
    nx = [((n >> s) & 1023) / 1023 * 2 - 1 for s in (0, 10, 20)]; print("  n", nx, "len2", sum(v * v for v in nx), "^512", sum(v * v for v in nx) ** 512)
