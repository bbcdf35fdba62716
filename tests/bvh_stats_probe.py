"""Manual probe: traversal statistics (node visits / leaf tests per ray, stack depth) of the frame's real
reflection rays, for the oracle's median-split BVH and -- on a GPU box -- for the product's LBVH.
python tests/bvh_stats_probe.py [W H mesh]"""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tests/", 1)[0])
sys.path.insert(0, __file__.rsplit("/", 1)[0])
import assets  # noqa: E402
from oracle import oracle as O  # noqa: E402


def stats(o, label):
    L = O.lib()
    o.set_threads(1)
    o.render_visibility()
    L.orc_tstats_reset()
    rays = o.ray_trace()
    out = (C.c_uint64 * 3)()
    L.orc_tstats_get(out)
    print("%-28s rays %d  node visits/ray %.1f  leaf tests/ray %.1f  max stack %d" % (label, rays, out[0] / rays, out[1] / rays, out[2]))


def main():
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 960
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 540
    mesh = sys.argv[3] if len(sys.argv) > 3 else "bunny.obj"
    o = O.Oracle(W, H)
    v, i, _ = O.obj_import(assets.path(mesh))
    o.set_mesh(1, v, i)
    o.set_env_dds(assets.path("rnl_cross.dds"))
    o.build_as()
    o.update_frame((10, 10, -24), O.camera_view_proj(W, H), 1 / 60)
    o.update_as()
    stats(o, "oracle median-split BVH")
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if have_gpu:
        from raytracedggx_amd import capi
        ctx = capi.Context(64, 64)
        ctx.set_mesh(1, v, i)
        ctx.build_as()
        for slot, (bn, bt) in enumerate(((capi.BUF_BVH_NODES0, capi.BUF_BVH_TRIS0), (capi.BUF_BVH_NODES1, capi.BUF_BVH_TRIS1))):
            o.set_bvh(slot, ctx.readback(bn), ctx.readback(bt), ctx.bvh_root(slot))
        stats(o, "product LBVH (30-bit Morton)")


if __name__ == "__main__":
    main()
