"""Manual probe (not a pytest file): renders frames with the HIP path and the oracle on the same
inputs and prints per-buffer agreement and per-pass timings.  python tests/gpu_probe.py [W H frames mesh]"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tests/", 1)[0])
sys.path.insert(0, __file__.rsplit("/", 1)[0])
import assets  # noqa: E402
from oracle import oracle as O  # noqa: E402
from raytracedggx_amd import app, capi  # noqa: E402


def rel_l2(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b ** 2).sum()), 1e-30))


def main():
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 640
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 360
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    mesh = sys.argv[4] if len(sys.argv) > 4 else "bunny.obj"
    metallic = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
    args = ["-mesh", assets.path(mesh), "-env", assets.path("rnl_cross.dds"), "-width", W, "-height", H]
    if metallic < 1.0:
        args += ["-metallic", metallic, metallic]
    t0 = time.time()
    a = app.RayTracedGGX(args)
    ctx = a.context
    print("init %.2fs" % (time.time() - t0))
    o = O.Oracle(W, H)
    v, i, _ = O.obj_import(assets.path(mesh))
    o.set_mesh(1, v, i)
    o.set_env_dds(assets.path("rnl_cross.dds"))
    if metallic < 1.0:
        o.set_metallic(0, metallic); o.set_metallic(1, metallic)
    # same BVH arrays on both sides
    for slot, (bn, bt) in enumerate(((capi.BUF_BVH_NODES0, capi.BUF_BVH_TRIS0), (capi.BUF_BVH_NODES1, capi.BUF_BVH_TRIS1))):
        o.set_bvh(slot, ctx.readback(bn), ctx.readback(bt), ctx.bvh_root(slot))
    # env decode parity
    _, _, env_o = o.env_texels()
    env_g = ctx.readback(capi.BUF_ENV)
    print("env texels equal:", np.array_equal(env_o, env_g), env_g.shape)
    o.transform_sh()
    ctx.enable_timing(1)
    vp = O.camera_view_proj(W, H)
    for f in range(frames):
        a.OnUpdate(); a.OnRender(); ctx.sync()
        o.set_frame_constants(a.frame_constants().tobytes()[:704] + o.get_frame_constants().tobytes()[704:])
        o.update_as(); o.render_visibility(); rays_o = o.ray_trace(); o.denoise(); o.tone_map()
        print("frame %d rays gpu %d oracle %d timings %s" % (f, ctx.ray_count(), rays_o, {k: round(v, 4) for k, v in ctx.timings().items()}))
        print("  tlas equal", np.array_equal(ctx.readback(capi.BUF_TLAS), o.inv_worlds()))
        if f == 0:
            print("  sh max abs diff", float(np.abs(ctx.readback(capi.BUF_SH_COEFFS) - o.buffer(O.BUF_SH_COEFFS)).max()))
        for name, gid, oid in (("visibility", capi.BUF_VISIBILITY, O.BUF_VISIBILITY), ("depth", capi.BUF_DEPTH, O.BUF_DEPTH),
                               ("normal", capi.BUF_NORMAL, O.BUF_NORMAL), ("roughMetal", capi.BUF_ROUGH_METAL, O.BUF_ROUGH_METAL),
                               ("velocity", capi.BUF_VELOCITY, O.BUF_VELOCITY), ("backbuffer", capi.BUF_BACKBUFFER, O.BUF_BACKBUFFER)):
            g, r = ctx.readback(gid), o.buffer(oid)
            print("  %-10s mismatching words: %d / %d" % (name, int((g != r).sum()), g.size))
            if name == "velocity" and (g != r).any():
                ys, xs = np.nonzero(g != r)
                for k in range(min(8, ys.size)):
                    gg, rr = g[ys[k], xs[k]], r[ys[k], xs[k]]
                    print("     (%d,%d) gpu %08x %s oracle %08x %s" % (xs[k], ys[k], gg, np.array([gg], np.uint32).view(np.float16), rr, np.array([rr], np.uint32).view(np.float16)))
        for name, gid, oid in (("rt_refl", capi.BUF_RT_REFL, O.BUF_RT_REFL), ("rt_diff", capi.BUF_RT_DIFF, O.BUF_RT_DIFF)):
            g, r = ctx.readback(gid), o.buffer(oid)
            print("  %-10s mismatching words: %d  relL2 %.3e" % (name, int((g != r).sum()), rel_l2(O.unpack_r11g11b10f(g), O.unpack_r11g11b10f(r))))
        p = ctx.frame_parity()
        assert p == o.parity()
        for name, gid, oid in (("flt_rfl", capi.BUF_FLT_RFL, O.BUF_FLT_RFL), ("flt_dff", capi.BUF_FLT_DFF, O.BUF_FLT_DFF),
                               ("tss[p]", capi.BUF_TSS0 + p, O.BUF_TSS0 + p)):
            g, r = ctx.readback(gid), o.buffer(oid)
            print("  %-10s mismatching words: %d  relL2 %.3e" % (name, int((g != r).sum()), rel_l2(O.unpack_rgba16f(g), O.unpack_rgba16f(r))))
    a.save_image("gpurun_out/probe.ppm")


if __name__ == "__main__":
    main()
