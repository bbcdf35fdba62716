"""Parity of the HIP path (through the C ABI / the C++ host layer) with the CPU oracle on identical
inputs.  Bar (BASELINE.json north_star): integer visibility / G-buffer words bit-exact; denoised HDR
within 1e-3 relative L2.  The tolerance exists because exp/exp2/log2/pow come from libm on the CPU and
from the device math library on the GPU; everything else is the same fp32 arithmetic, unfused."""
import os

import numpy as np
import pytest

import assets
import bvh_checks
from oracle import oracle as O

pytestmark = pytest.mark.gpu

HDR_TOL = 1e-3   # relative L2, from north_star
# ... against the oracle's EXACT evaluation of the filters' normal weight pow(dot(N, Nc), 512).  Against its plain-fp32 reading of the HLSL
# ("libm": fp32 dot product, std::pow in fp32) the bar is wider, and stated here rather than met by sharing a rounding: ONE fp32 rounding of
# a dot product next to 1 is 6e-8, times 512 in the weight = 3e-5 -- what any two faithful fp32 implementations of the shader differ by
# (HLSL fixes neither the order of a dp3 nor the last bits of pow) --, and the temporal pass turns a 1e-5 difference of the filtered image into
# 1e-3 of its result (measured: 1.03e-3 on the 1080p bunny, frame 1).  The product evaluates the exact value to a few ulps.
HDR_TOL_FP32_ORACLE = 2e-3


class _DeviceView:
    """A raw device pointer through __cuda_array_interface__ (torch wraps it without a copy)."""
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 3}


def rel_l2(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b ** 2).sum()), 1e-30))


class Pair:
    """The product's RayTracedGGX application object and an oracle on the same scene."""

    def __init__(self, W, H, mesh="bunny.obj", metallic=None, pos_scale=None, env_const=None, shared_mem=False, normal_weight="exact"):
        from raytracedggx_amd import app, capi
        self.capi = capi
        args = ["-mesh", assets.path(mesh)] + ([str(x) for x in pos_scale] if pos_scale else []) + \
               ["-env", assets.path("rnl_cross.dds"), "-width", W, "-height", H]
        if metallic is not None:
            args += ["-metallic", metallic[0], metallic[1]]
        if shared_mem:
            args += ["-sharedmem"]          # the [V] toggle of the sample: LDS-staged spatial filters
        self.app = app.RayTracedGGX(args)
        self.ctx = self.app.context
        self.o = O.Oracle(W, H)
        # how the oracle evaluates the filters' pow(dot(N, Nc), 512): "exact" (double, rounded once) or "libm" (a plain fp32 reading of the
        # HLSL) -- neither shares a rounding with the product's v_exp_f32(512 v_log_f32 x) (oracle/orc_denoise.h normal_weight; process-wide)
        self.o.set_normal_weight(normal_weight)
        v, i, _ = O.obj_import(assets.path(mesh))
        self.o.set_mesh(1, v, i)
        if pos_scale:
            self.o.set_pos_scale(pos_scale)
        if env_const is not None:
            env = assets.constant_env_rgba16f(env_const)
            self.ctx.set_env(capi.FORMAT_RGBA16F, 1, 1, env)
            self.o.set_env_rgba16f(1, 1, env)
        else:
            self.o.set_env_dds(assets.path("rnl_cross.dds"))
        if metallic is not None:
            self.o.set_metallic(0, metallic[0]); self.o.set_metallic(1, metallic[1])
        self.hdr_tol = HDR_TOL if normal_weight == "exact" else HDR_TOL_FP32_ORACLE
        self.num_tris = [12, i.size // 3]
        self.give_oracle_the_device_trees()
        self.o.transform_sh()
        self.rays = None

    def give_oracle_the_device_trees(self, refitted=False):
        """The CPU re-traces the same BVH arrays the HIP kernels use -- after checking them: every primitive in exactly one
        leaf, every box tight around what is below it, the 4-wide collapse equal to the binary tree and chosen by the surface-area
        rule (tests/bvh_checks.py; refitted: the model's tree keeps the choice its build made for another shape)."""
        capi = self.capi
        for slot, (bn, bt, b4, btop, cap) in enumerate(((capi.BUF_BVH_NODES0, capi.BUF_BVH_TRIS0, capi.BUF_BVH4_NODES0, capi.BUF_BVH4_TOP0, 16),
                                                        (capi.BUF_BVH_NODES1, capi.BUF_BVH_TRIS1, capi.BUF_BVH4_NODES1, capi.BUF_BVH4_TOP1, 96))):
            nodes, tris, root = self.ctx.readback(bn), self.ctx.readback(bt), self.ctx.bvh_root(slot)
            bvh_checks.bvh_check(nodes, tris, root, self.num_tris[slot])
            nodes4 = self.ctx.readback(b4)
            bvh_checks.bvh4_check(nodes, nodes4, root, built_shape=not (refitted and slot == 1), weights=self.ctx.collapse_weights())
            bvh_checks.bvh4_top_check(nodes4, self.ctx.readback(btop), root, cap)
            self.o.set_bvh(slot, nodes, tris, root)

    def frame(self):
        self.app.OnUpdate(); self.app.OnRender(); self.ctx.sync()
        # the oracle consumes the constants the product's host layer produced (its own are checked in test_host_and_abi)
        self.o.set_frame_constants(self.app.frame_constants().tobytes()[:704] + self.o.get_frame_constants().tobytes()[704:])
        self.o.update_as(); self.o.render_visibility(); self.rays = self.o.ray_trace(); self.o.denoise(); self.o.tone_map()

    def close(self):
        self.app.OnDestroy(); self.o.close()

    def check_frame(self, label):
        capi, ctx, o = self.capi, self.ctx, self.o
        for name, gid, oid in (("visibility", capi.BUF_VISIBILITY, O.BUF_VISIBILITY), ("depth", capi.BUF_DEPTH, O.BUF_DEPTH),
                               ("normal", capi.BUF_NORMAL, O.BUF_NORMAL), ("roughMetal", capi.BUF_ROUGH_METAL, O.BUF_ROUGH_METAL),
                               ("velocity", capi.BUF_VELOCITY, O.BUF_VELOCITY)):
            np.testing.assert_array_equal(ctx.readback(gid), o.buffer(oid), err_msg="%s: %s not bit-exact" % (label, name))
        np.testing.assert_array_equal(ctx.readback(capi.BUF_TLAS), o.inv_worlds())
        assert ctx.ray_count() == self.rays, "%s: ray count" % label
        # the raw traced images: bit for bit since round 3 (the shading path's exp2 / log2 are the numeric contract's on both sides,
        # rtggx_device.h exp2Contract / log2Contract; until then a word could differ by one code)
        for name, gid, oid in (("rt_refl", capi.BUF_RT_REFL, O.BUF_RT_REFL), ("rt_diff", capi.BUF_RT_DIFF, O.BUF_RT_DIFF)):
            np.testing.assert_array_equal(ctx.readback(gid), o.buffer(oid), err_msg="%s: %s not bit-exact" % (label, name))
        p = ctx.frame_parity()
        assert p == o.parity()
        for name, gid, oid in (("FilteredOut", capi.BUF_FLT_RFL, O.BUF_FLT_RFL), ("FilteredOut1", capi.BUF_FLT_DFF, O.BUF_FLT_DFF),
                               ("TemporalSSOut", capi.BUF_TSS0 + p, O.BUF_TSS0 + p)):
            g, r = O.unpack_rgba16f(ctx.readback(gid)), O.unpack_rgba16f(o.buffer(oid))
            # NaNs are part of the reference's behaviour near the frame border (0 x inf in ReflectionWeight for taps that
            # read outside the image, SpatialFilter.hlsli:60): they must appear in the same pixels, nowhere else
            fin = np.isfinite(r)
            np.testing.assert_array_equal(np.isfinite(g), fin, err_msg="%s: %s non-finite values differ from the oracle's" % (label, name))
            e = rel_l2(np.where(fin, g, 0.0), np.where(fin, r, 0.0))
            assert e < self.hdr_tol, "%s: %s relative L2 %.3e" % (label, name, e)
        g, r = O.unpack_rgba8(ctx.readback(capi.BUF_BACKBUFFER)).astype(int), O.unpack_rgba8(o.buffer(O.BUF_BACKBUFFER)).astype(int)
        # 8-bit codes: a value on a rounding boundary may land on either side (the denoiser uses v_rcp/v_sqrt where the
        # oracle divides); never more than one code, rarely, and far inside the 1e-3 bar as an image
        assert np.abs(g - r).max() <= 1 and (g != r).mean() < 2e-2 and rel_l2(g, r) < HDR_TOL, "%s: back buffer" % label


def test_config_c1_single_triangle_constant_env(built):
    p = Pair(256, 256, mesh="triangle.obj", env_const=1.0)
    try:
        for f in range(2):
            p.frame(); p.check_frame("C1 frame %d" % f)
        assert p.rays > 0
    finally:
        p.close()


@pytest.mark.parametrize("shared_mem", [False, True], ids=["direct", "sharedmem"])
def test_bunny_three_frames(built, shared_mem):
    p = Pair(640, 360, shared_mem=shared_mem)
    try:
        np.testing.assert_array_equal(p.ctx.readback(p.capi.BUF_ENV), p.o.env_texels()[2])          # BC6H decode: bit-exact
        for f in range(3):
            p.frame(); p.check_frame("bunny frame %d" % f)
        np.testing.assert_allclose(p.ctx.readback(p.capi.BUF_SH_COEFFS), p.o.buffer(O.BUF_SH_COEFFS), rtol=1e-5, atol=1e-6)
        assert p.rays > 50000
    finally:
        p.close()


@pytest.mark.parametrize("mesh,tris", [("bunny.obj", 69666), ("dragon.obj", 100000)])
def test_wide_tree_filled_by_surface_area(built, mesh, tris):
    """The 4-wide collapse (round 4; the reference asks its driver for PREFER_FAST_TRACE, RayTracer.cpp:676-716): every 4-wide node holds
    the entries the surface-area rule gives it (bvh_checks.bvh4_check re-derives them from the binary tree), the tree is at least 3.5
    entries per node full (the depth-parity collapse of rounds 1-3: 3.0), and rays against the oracle's walk of the BINARY tree agree
    bit for bit."""
    p = Pair(480, 270, mesh=mesh, metallic=(1.0, 0.5), shared_mem=True)
    try:
        n4 = p.ctx.readback(p.capi.BUF_BVH4_NODES1).reshape(-1, 32)
        used = n4.any(axis=1)
        refs = n4[used, 24:28].view(np.int32)
        entries = int((refs != 0x7FFFFFFF).sum())
        nodes = int(used.sum())
        assert entries == tris + nodes - 1, "every triangle and every node but the root is an entry once"
        fill = entries / nodes
        print("%s: %d 4-wide nodes for %d triangles, %.2f entries per node, %d levels" % (mesh, nodes, tris, fill, int(n4[used, 28].max()) + 1))
        assert fill >= 3.4, "%s: %.2f entries per 4-wide node" % (mesh, fill)
        for f in range(2):
            p.frame(); p.check_frame("%s frame %d" % (mesh, f))
        rng = np.random.default_rng(11)
        n = 8000
        org = np.array([10.0, 10.0, -24.0]) + rng.standard_normal((n, 3)) * 2.0
        tgt = np.stack([rng.uniform(-8, 8, n), rng.uniform(-1, 10, n), rng.uniform(-8, 8, n)], 1)
        rays = np.concatenate([org, tgt - org, np.full((n, 1), 1e-5), np.full((n, 1), 1e4)], 1).astype(np.float32)
        g, c = p.ctx.trace_rays(rays), p.o.trace_rays(rays)      # the oracle walks the BINARY tree
        for k in ("valid", "inst", "prim", "t", "b1", "b2"):
            sel = slice(None) if k == "valid" else c["valid"]
            np.testing.assert_array_equal(g[k][sel], c[k][sel], err_msg=k)
        assert g["valid"].sum() > n // 4
    finally:
        p.close()


def test_vndf_sampler_against_its_oracle_counterpart(built):
    """The opt-in sampler (rtggx_set_sampler / -vndf; north_star: "GGX-VNDF importance sampling"): visible-normal sampling of the reflection
    lobe (Heitz 2018) in rayGenKernel against its restatement in the oracle -- the whole frame check, three frames, rays bit for bit.
    It is a different estimator from the reference's NDF sampling (the default, every other test): the traced image differs, the
    ray count differs (no sample below the view's horizon), and the mean radiance agrees (both are unbiased estimates of the same lobe)."""
    p = Pair(480, 270, metallic=(1.0, 1.0), shared_mem=True)
    q = Pair(480, 270, metallic=(1.0, 1.0), shared_mem=True)
    try:
        p.ctx.set_sampler(True); p.o.set_sampler(True)
        for f in range(3):
            p.frame(); p.check_frame("vndf frame %d" % f)
            q.frame()
        a, b = O.unpack_r11g11b10f(p.ctx.readback(p.capi.BUF_RT_REFL)).astype(np.float64), O.unpack_r11g11b10f(q.ctx.readback(q.capi.BUF_RT_REFL)).astype(np.float64)
        covered = p.ctx.readback(p.capi.BUF_VISIBILITY) != 0
        assert (np.abs(a - b).sum(axis=-1)[covered] > 0).mean() > 0.5, "a different sampler: most covered pixels trace another direction"
        assert p.rays >= q.rays, "visible normals never reflect below the surface where the plain distribution does (%d vs %d rays)" % (p.rays, q.rays)
        ma, mb = np.median(a[covered], axis=0), np.median(b[covered], axis=0)
        assert np.all(np.abs(ma - mb) < 0.35 * np.maximum(mb, 1e-3)), "the two estimators agree on the typical radiance (%s vs %s)" % (ma, mb)
    finally:
        p.close(); q.close()


def test_dragon_diffuse_path(built):
    # metallic < 1 on both meshes: second ray per pixel, SH irradiance, both closest-hit groups (RayTracing.hlsl:559-564, 593-614)
    p = Pair(480, 270, mesh="dragon.obj", metallic=(0.25, 0.5), shared_mem=True)
    try:
        for f in range(2):
            p.frame(); p.check_frame("dragon metallic<1 frame %d" % f)
        vis = p.o.buffer(O.BUF_VISIBILITY)
        assert p.rays > 1.5 * (vis > 0).sum()
    finally:
        p.close()


def test_turing_bowl_style_placement(built):
    # Bin/TuringBowl.bat passes position and scale; exercised here with the bunny: -mesh <obj> 0 2.8 0 0.3
    p = Pair(320, 180, pos_scale=(0.0, 2.8, 0.0, 0.3))
    try:
        p.frame(); p.check_frame("placed bunny")
    finally:
        p.close()


@pytest.mark.parametrize("size", [(333, 217), (97, 61), (16, 16), (1, 1)], ids=lambda s: "%dx%d" % s)
def test_ragged_frame_sizes(built, size):
    """Widths and heights that are no multiple of any tile size (16x16 ray tiles, 64x4 / 16x16 filter blocks, 64-pixel
    raster tiles), down to a single pixel: every pass must clip its aprons and partial tiles exactly."""
    p = Pair(size[0], size[1], metallic=(0.5, 0.25), shared_mem=True)
    try:
        for f in range(2):
            p.frame(); p.check_frame("%dx%d frame %d" % (size[0], size[1], f))
    finally:
        p.close()


def test_model_off_screen_and_behind_camera(built):
    """Nothing of the model is visible (placed far behind the camera): triangles with w <= 0 are dropped, the frame is
    ground + environment only, and the ray tracer still sees the model through reflections."""
    p = Pair(320, 180, pos_scale=(0.0, 0.0, -400.0, 1.0))
    try:
        p.frame(); p.check_frame("model behind the camera")
        vis = p.ctx.readback(p.capi.BUF_VISIBILITY)
        assert (vis >= 0x01000000).sum() == 0 and (vis > 0).sum() > 0, "only the ground is visible"
    finally:
        p.close()


def test_model_crossing_the_near_plane(built):
    """The model stands right beside the camera and reaches behind it: triangles cross z_clip = 0 and w = 0 and are clipped against the
    near plane (one or two sub-triangles with the primitive's id) instead of being dropped -- same words as the oracle."""
    p = Pair(320, 180, pos_scale=(14.8, 4.5, -19.9, 1.0), shared_mem=True)     # just to the right of the eye, reaching behind it
    try:
        p.frame(); p.check_frame("model around the camera")
        vis = p.ctx.readback(p.capi.BUF_VISIBILITY)
        depth = p.ctx.readback(p.capi.BUF_DEPTH)
        model = vis >= 0x01000001
        assert model.mean() > 0.05, "the model covers part of the frame"
        assert (depth[model] < 16777215 // 2).any(), "with fragments close to the near plane"
    finally:
        p.close()


def test_triangle_far_beyond_the_guard_band(built):
    """Round 4: a triangle whose vertices lie 10^7 units to the sides of the view -- 10^5 viewport widths, far beyond the 2^30 sub-pixel
    units the snapped coordinates may take -- is clipped against the guard band (|x|, |y| <= 256 w in clip space: oracle/orc_raster.h
    holds the contract) and rasterised as a fan, where rounds 1-3 dropped it: it covers the frame, with the oracle's words and depths,
    and the rest of the frame's check holds on it."""
    p = Pair(320, 180, mesh="triangle.obj", pos_scale=(0.0, -1.0e6, 0.0, 1.0e7), shared_mem=True)
    try:
        p.frame(); p.check_frame("a triangle 2 x 10^7 units wide")
        vis = p.ctx.readback(p.capi.BUF_VISIBILITY)
        assert (vis == 0x01000001).mean() > 0.7, "the triangle covers the frame where the ground does not hide it: %.3f" % (vis == 0x01000001).mean()
    finally:
        p.close()


def test_lbvh_structure_and_device_traversal(built):
    from raytracedggx_amd import capi
    p = Pair(64, 64)
    try:
        p.frame()
        depth = bvh_checks.bvh_check(p.ctx.readback(capi.BUF_BVH_NODES1), p.ctx.readback(capi.BUF_BVH_TRIS1), p.ctx.bvh_root(1), 69666)
        assert depth <= 48, "unexpectedly deep tree for 69666 triangles: %d levels" % depth
        bvh_checks.bvh_check(p.ctx.readback(capi.BUF_BVH_NODES0), p.ctx.readback(capi.BUF_BVH_TRIS0), p.ctx.bvh_root(0), 12)
        used = bvh_checks.bvh4_check(p.ctx.readback(capi.BUF_BVH_NODES1), p.ctx.readback(capi.BUF_BVH4_NODES1), p.ctx.bvh_root(1), weights=p.ctx.collapse_weights())
        assert 69665 // 16 <= used <= 69665, "4-wide nodes: %d" % used      # (multi-leaves of up to four triangles: ~10 000 nodes; with single-triangle leaves ~35 000)
        bvh_checks.bvh4_check(p.ctx.readback(capi.BUF_BVH_NODES0), p.ctx.readback(capi.BUF_BVH4_NODES0), p.ctx.bvh_root(0), weights=p.ctx.collapse_weights())
        rng = np.random.default_rng(11)
        n = 20000
        org = np.array([10.0, 10.0, -24.0]) + rng.standard_normal((n, 3)) * 2.0
        tgt = np.stack([rng.uniform(-8, 8, n), rng.uniform(-1, 10, n), rng.uniform(-8, 8, n)], 1)
        rays = np.concatenate([org, tgt - org, np.full((n, 1), 1e-5), np.full((n, 1), 1e4)], 1).astype(np.float32)
        rays[::97, 6:] = 0.0                                                 # degenerate intervals never hit
        g = p.ctx.trace_rays(rays)
        c = p.o.trace_rays(rays)                                             # same BVH arrays, CPU traversal
        for k in ("valid", "inst", "prim", "t", "b1", "b2"):
            sel = slice(None) if k == "valid" else c["valid"]
            np.testing.assert_array_equal(g[k][sel], c[k][sel], err_msg=k)
        b = p.o.trace_rays(rays[:1500], brute=True)                          # and no BVH at all
        np.testing.assert_array_equal(g["valid"][:1500], b["valid"])
        hit = b["valid"]
        np.testing.assert_array_equal(g["prim"][:1500][hit], b["prim"][hit])
        np.testing.assert_array_equal(g["t"][:1500][hit], b["t"][hit])
        assert not g["valid"][::97].any() and g["valid"].sum() > n // 4
    finally:
        p.close()


@pytest.mark.parametrize("shared_mem", [False, True], ids=["direct", "sharedmem"])
def test_denoiser_in_isolation_on_uploaded_inputs(built, shared_mem):
    """Feed the oracle's G-buffer and raw ray-traced images to rtggx_denoise: isolates the filter chain."""
    from raytracedggx_amd import capi
    W, H = 384, 216
    p = Pair(W, H, metallic=(0.5, 0.75))
    try:
        p.frame()   # gives both sides constants, parity and history
        ctx, o = p.ctx, p.o
        # second frame: trace on the CPU only, upload, denoise on the GPU
        p.app.OnUpdate()
        o.set_frame_constants(p.app.frame_constants().tobytes()[:704] + o.get_frame_constants().tobytes()[704:])
        o.update_as(); o.render_visibility(); o.ray_trace()
        for gid, oid in ((capi.BUF_VISIBILITY, O.BUF_VISIBILITY), (capi.BUF_DEPTH, O.BUF_DEPTH), (capi.BUF_NORMAL, O.BUF_NORMAL),
                         (capi.BUF_ROUGH_METAL, O.BUF_ROUGH_METAL), (capi.BUF_VELOCITY, O.BUF_VELOCITY), (capi.BUF_RT_REFL, O.BUF_RT_REFL),
                         (capi.BUF_RT_DIFF, O.BUF_RT_DIFF), (capi.BUF_TSS0, O.BUF_TSS0), (capi.BUF_TSS1, O.BUF_TSS1)):
            ctx.upload(gid, o.buffer(oid))
        ctx.denoise(shared_mem); ctx.tone_map(); ctx.sync()
        o.denoise(); o.tone_map()
        par = ctx.frame_parity()
        assert par == o.parity()
        for gid, oid in ((capi.BUF_FLT_RFL, O.BUF_FLT_RFL), (capi.BUF_FLT_DFF, O.BUF_FLT_DFF), (capi.BUF_TSS0 + par, O.BUF_TSS0 + par)):
            assert rel_l2(O.unpack_rgba16f(ctx.readback(gid)), O.unpack_rgba16f(o.buffer(oid))) < HDR_TOL
        # alpha channels are exact: hit flag (FilteredOut.w), history weight quantised to 4 bits/15
        np.testing.assert_array_equal(ctx.readback(capi.BUF_FLT_RFL) >> np.uint64(48), o.buffer(O.BUF_FLT_RFL) >> np.uint64(48))
    finally:
        p.close()


def _rays_against_an_independent_tree(p, mesh, pos_scale=None, n=20000, seed=11):
    """Closest hits of random rays: the HIP traversal of the DEVICE-built tree against the oracle walking a tree of ITS OWN
    (oracle/orc_bvh.h, built on the CPU from the mesh).  The closest hit is a property of the ray and the triangle set, not of
    the hierarchy, so instance, primitive, t and barycentrics must agree bit for bit -- a check the device builder cannot
    influence (every other parity test hands the oracle the device's arrays)."""
    o2 = O.Oracle(64, 64)
    try:
        v, i, aabb = O.obj_import(assets.path(mesh))
        o2.set_mesh(1, v, i)
        if pos_scale:
            o2.set_pos_scale(pos_scale)
        o2.build_as()
        o2.set_frame_constants(p.app.frame_constants().tobytes()[:704] + o2.get_frame_constants().tobytes()[704:])
        o2.update_as()
        ps = np.asarray(pos_scale if pos_scale else (0, 0, 0, 1), np.float64)
        lo, hi = aabb[:3] * ps[3] + ps[:3], aabb[3:] * ps[3] + ps[:3]
        rng = np.random.default_rng(seed)
        org = np.array([10.0, 10.0, -24.0]) + rng.standard_normal((n, 3)) * 2.0
        tgt = rng.uniform(np.minimum(lo, hi) - 1.0, np.maximum(lo, hi) + 1.0, (n, 3))
        rays = np.concatenate([org, tgt - org, np.full((n, 1), 1e-5), np.full((n, 1), 1e4)], 1).astype(np.float32)
        g, c = p.ctx.trace_rays(rays), o2.trace_rays(rays)
        np.testing.assert_array_equal(g["valid"], c["valid"])
        for k in ("inst", "prim", "t", "b1", "b2"):
            np.testing.assert_array_equal(g[k][c["valid"]], c[k][c["valid"]], err_msg="%s: %s" % (mesh, k))
        assert (g["inst"][g["valid"]] == 1).sum() > n // 8, "a good share of the rays hit the model"
    finally:
        o2.close()


def _full_size_properties(W, H, mesh, label, checked_frames=0, normal_weight="exact"):
    """A BASELINE.json configuration at its full size.  `checked_frames` frames with the WHOLE parity check of the small cases
    (Pair.check_frame: integer buffers bit-exact, raw traced words within one code, FilteredOut / FilteredOut1 / the temporal result
    with its history alpha inside the bar, the back buffer within one code) -- from the second frame on the history is in play --,
    then one more frame: integer buffers and the ray count against the oracle, the denoised image inside the bar, determinism,
    strip independence (the two halves rendered separately give the same words)."""
    from raytracedggx_amd import capi
    p = Pair(W, H, mesh=mesh, shared_mem=True, normal_weight=normal_weight)
    try:
        for f in range(checked_frames):
            p.frame(); p.check_frame("%s frame %d" % (label, f))
        p.frame()
        ctx, o = p.ctx, p.o
        for gid, oid in ((capi.BUF_VISIBILITY, O.BUF_VISIBILITY), (capi.BUF_DEPTH, O.BUF_DEPTH), (capi.BUF_NORMAL, O.BUF_NORMAL),
                         (capi.BUF_ROUGH_METAL, O.BUF_ROUGH_METAL), (capi.BUF_VELOCITY, O.BUF_VELOCITY)):
            np.testing.assert_array_equal(ctx.readback(gid), o.buffer(oid), err_msg="%s buffer %d" % (label, gid))
        assert ctx.ray_count() == p.rays
        tss = O.unpack_rgba16f(ctx.readback(capi.BUF_TSS0 + ctx.frame_parity()))
        assert rel_l2(tss, O.unpack_rgba16f(o.buffer(O.BUF_TSS0 + o.parity()))) < p.hdr_tol
        g, r = ctx.readback(capi.BUF_BACKBUFFER), o.buffer(O.BUF_BACKBUFFER)
        # 8-bit codes.  With a history in play (checked_frames > 0) the temporal pass's clamp window -- gamma <= 32 times the square root of a
        # 3x3 variance that is rounding noise where the image is flat -- turns the 1e-5 by which the filtered inputs differ into a percent on
        # a handful of pixels (tools/probes/parity_probe.py): at two million pixels a frame, one of them now and then lands two codes away.
        d = np.abs(O.unpack_rgba8(g).astype(int) - O.unpack_rgba8(r).astype(int))
        assert d.max() <= (2 if checked_frames else 1) and (d > 1).mean() < 1e-5, "%s: back buffer differs by up to %d codes, %d values by more than one" % (label, d.max(), (d > 1).sum())
        vis = ctx.readback(capi.BUF_VISIBILITY)
        # coverage sanity: model + slab cover 15-50 % of the frame, the model is in front of the slab somewhere
        assert 0.15 < (vis > 0).mean() < 0.5 and (vis >= 0x01000000).mean() > 0.03
        # determinism: re-issuing the same passes (same constants) reproduces every integer buffer and the raw reflection image
        before = {b: ctx.readback(b) for b in (capi.BUF_VISIBILITY, capi.BUF_DEPTH, capi.BUF_NORMAL, capi.BUF_VELOCITY, capi.BUF_RT_REFL)}
        ctx.render_visibility(); ctx.ray_trace(); ctx.sync()
        for b, ref in before.items():
            np.testing.assert_array_equal(ctx.readback(b), ref)
        # strip independence (multi-GPU tiling): rendering the two halves separately gives the same
        # words in those rows (the full pipeline with history exchange: the strip tests below)
        for r0, r1 in ((0, H // 2), (H // 2, H)):
            ctx.set_strip(r0, r1)
            ctx.update_frame(p.app.frame_constants())
            ctx.update_as(); ctx.render_visibility(); ctx.ray_trace(); ctx.sync()
            for b, ref in before.items():
                np.testing.assert_array_equal(ctx.readback(b)[r0:r1], ref[r0:r1])
        ctx.set_strip(0, H)
        return p
    except BaseException:
        p.close()
        raise


@pytest.mark.parametrize("normal_weight", ["exact", "libm"])
def test_full_size_1080p_properties(built, normal_weight):
    """BASELINE.json configs[1] (the bench workload) at full size: bunny 1920x1080, all-metal -- three frames with the whole parity check,
    against BOTH independent evaluations of the filters' 512th power in the oracle (round 3's check held because product and oracle
    shared nine squarings and their 400 ulps; VERDICT r03 "weak" 2)."""
    _full_size_properties(1920, 1080, "bunny.obj", "C2 (%s power)" % normal_weight, checked_frames=3, normal_weight=normal_weight).close()


@pytest.mark.parametrize("normal_weight", ["exact", "libm"])
def test_c3_dragon_1080p_all_metal(built, normal_weight):
    """BASELINE.json configs[2]: the dragon at 1920x1080 with the default all-metal materials (stpeters_cross.dds is not in the
    reference tree: rnl_cross.dds stands in) -- the full-size property set, the dragon tree structure-checked
    (Pair), and the device traversal of the device tree against the oracle walking a tree of its own."""
    p = _full_size_properties(1920, 1080, "dragon.obj", "C3 (%s power)" % normal_weight, checked_frames=3, normal_weight=normal_weight)
    try:
        assert p.num_tris[1] == 100000
        if normal_weight == "exact":
            _rays_against_an_independent_tree(p, "dragon.obj")
    finally:
        p.close()


def test_turing_bowl_scene(built):
    """Bin/TuringBowl.bat: `-mesh Assets/TuringBowl.obj 0.0 2.8 0.0 0.03` -- the one shipped mesh with `vn` records (per-corner
    normals, vertex splitting: XUSGObjLoader.cpp:300-335), placed and scaled by the command line.  Frames against the oracle,
    tree structure-checked, traversal against an independent tree."""
    ps = (0.0, 2.8, 0.0, 0.03)
    p = Pair(640, 360, mesh="TuringBowl.obj", pos_scale=ps, shared_mem=True)
    try:
        assert p.num_tris[1] == 22744
        for f in range(2):
            p.frame(); p.check_frame("TuringBowl frame %d" % f)
        vis = p.ctx.readback(p.capi.BUF_VISIBILITY)
        assert (vis >= 0x01000000).mean() > 0.05, "the bowl is on screen"
        _rays_against_an_independent_tree(p, "TuringBowl.obj", pos_scale=ps)
    finally:
        p.close()


def test_two_strips_with_history_exchange_equal_one_frame(built):
    """two row strips (two contexts, the per-frame exchange plan carried out by copies) against one
    context rendering the whole frame -- every frame's back buffer and temporal history are bit-identical."""
    from raytracedggx_amd import capi
    from raytracedggx_amd.strips import StripRenderer
    W, H = 640, 360
    mesh, env = assets.path("bunny.obj"), assets.path("rnl_cross.dds")
    strips = []

    def transport(r, plan):
        for op, name, r0, r1, peer in plan:
            if op != "recv":
                continue
            bid = capi.BUF_BACKBUFFER if name == "backbuffer" else capi.BUF_TSS0 + r.context.frame_parity()
            mine = r.context.readback(bid)
            mine[r0:r1] = strips[peer].context.readback(bid)[r0:r1]
            r.context.upload(bid, mine)

    full = StripRenderer(W, H, mesh, env, extra_args=("-sharedmem",))
    strips += [StripRenderer(W, H, mesh, env, rank=r, world=2, transport=transport, extra_args=("-sharedmem",)) for r in range(2)]
    try:
        rays = 0
        for f in range(4):
            full.frame(); full.context.sync()
            for s in strips:
                s.render(); s.context.sync()
            for s in strips:
                s.exchange()
            par = full.context.frame_parity()
            np.testing.assert_array_equal(strips[0].context.readback(capi.BUF_BACKBUFFER), full.context.readback(capi.BUF_BACKBUFFER),
                                          err_msg="frame %d: assembled back buffer" % f)
            ref = full.context.readback(capi.BUF_TSS0 + par)
            for s in strips:
                assert s.context.frame_parity() == par
                np.testing.assert_array_equal(s.context.readback(capi.BUF_TSS0 + par)[s.b:s.e], ref[s.b:s.e], err_msg="frame %d: history" % f)
            assert sum(s.context.ray_count() for s in strips) == full.context.ray_count(), "rays are counted once, by the strip that owns the pixel"
            rays += full.context.ray_count()
        assert rays > 100000
    finally:
        full.close()
        for s in strips:
            s.close()


def test_strips_through_torch_views_on_torch_stream(built):
    """The multi-GPU code path minus RCCL: the exchanged targets wrapped as torch tensors (zero-copy views of the
    library's device buffers), all passes on torch's current stream, rows moved with torch copies on that stream --
    against the single-context frame.  Checks the views alias the right memory and that stream ordering holds."""
    import torch
    from raytracedggx_amd import capi
    from raytracedggx_amd.strips import StripRenderer
    W, H = 480, 272
    mesh, env = assets.path("bunny.obj"), assets.path("rnl_cross.dds")
    strips = []

    def transport(r, plan):
        mine = r.exchange_buffers()
        with torch.cuda.stream(r.stream):
            for op, name, r0, r1, peer in plan:
                if op == "recv":
                    r.stream.wait_stream(strips[peer].stream)       # what RCCL's send/recv pairing does across processes
                    mine[name][r0:r1].copy_(strips[peer].exchange_buffers()[name][r0:r1], non_blocking=True)

    full = StripRenderer(W, H, mesh, env, extra_args=("-sharedmem",))
    strips += [StripRenderer(W, H, mesh, env, rank=r, world=2, transport=transport, torch_buffers=True, extra_args=("-sharedmem",)) for r in range(2)]
    try:
        for f in range(3):
            full.frame()
            for s in strips:
                s.render()
            for s in strips:
                s.exchange()
            for s in strips:                                         # a peer may not start its next frame (and overwrite
                for t in strips:                                     # rows being copied) before the copies are done
                    s.stream.wait_stream(t.stream)
            torch.cuda.synchronize(); full.context.sync()
            assert strips[0].exchange_buffers()["backbuffer"].data_ptr() == strips[0].context.buffer_ptr(capi.BUF_BACKBUFFER)
            np.testing.assert_array_equal(strips[0].exchange_buffers()["backbuffer"].cpu().numpy().view(np.uint32),
                                          full.context.readback(capi.BUF_BACKBUFFER), err_msg="frame %d" % f)
            np.testing.assert_array_equal(strips[0].context.readback(capi.BUF_BACKBUFFER), full.context.readback(capi.BUF_BACKBUFFER))
    finally:
        full.close()
        for s in strips:
            s.close()


def test_adaptive_split_of_expensive_bins_changes_nothing(built):
    """Bins that were expensive in the previous frame are traced by 2, 4 or 8 waves in the next (trace.hip "adaptive
    split"; hits merge with a 64-bit atomic min).  Five frames at 1280x720 with the split forced hard (a new wave per 40
    lane-steps, up to 8 per bin) against the same frames with it off: every target bit-identical, and the split list in use."""
    from raytracedggx_amd import app, capi
    args = ["-mesh", assets.path("bunny.obj"), "-env", assets.path("rnl_cross.dds"), "-width", 1280, "-height", 720, "-sharedmem", "-metallic", 0.5, 0.5]
    a, b = app.RayTracedGGX(args), app.RayTracedGGX(args)
    try:
        a.context.debug_trace_split(0, 0)
        b.context.debug_trace_split(40, 3, 16384)
        demands = []
        for f in range(5):
            for x in (a, b):
                x.OnUpdate(); x.OnRender(); x.context.sync()
            demands.append(b.context.debug_trace_split(40, 3, 16384))
            assert a.context.debug_trace_split(0, 0) == 0
            for bid in (capi.BUF_VISIBILITY, capi.BUF_DEPTH, capi.BUF_NORMAL, capi.BUF_RT_REFL, capi.BUF_RT_DIFF, capi.BUF_TSS0, capi.BUF_TSS1, capi.BUF_BACKBUFFER):
                np.testing.assert_array_equal(a.context.readback(bid), b.context.readback(bid), err_msg="frame %d buffer %d" % (f, bid))
        # the first frame knows no costs yet -- nor the second: ray generation of frame f reads what the traversal of frame f - 2
        # recorded (frame f - 1's may still be running beside it: the cost record exists twice, by frame parity)
        assert demands[0] == 0 and min(demands[2:]) > 1000, demands
    finally:
        a.OnDestroy(); b.OnDestroy()


def test_free_running_frames_equal_synchronised_frames(built):
    """The host may run ahead of the GPU (three input sets, four constant slots, the fence in rtggx_render_visibility;
    the ray counters and the split-list demand come back asynchronously): 40 frames issued without a single
    synchronisation against the same 40 frames with a sync after each -- every target bit-identical at the end, at a
    size where the trace launch uses one wave per bin and the adaptive split (1280x720), at a thin one (1920x64) and at the
    full all-metal frame (three streams at work)."""
    from raytracedggx_amd import app, capi
    for size, metallic in (((1280, 720), (1.0, 0.5)), ((1920, 64), (1.0, 0.5)), ((1920, 1080), (1.0, 1.0))):      # the last: all-metal, where the
        # visibility pass of the next frame runs on its own stream beside a traversal that uses the adaptive split
        args = ["-mesh", assets.path("bunny.obj"), "-env", assets.path("rnl_cross.dds"), "-width", size[0], "-height", size[1], "-sharedmem", "-metallic", metallic[0], metallic[1]]
        a, b = app.RayTracedGGX(args), app.RayTracedGGX(args)
        try:
            for f in range(40):
                a.OnUpdate(); a.OnRender(); a.context.sync()
                b.OnUpdate(); b.OnRender()
            b.context.sync()
            for bid in (capi.BUF_VISIBILITY, capi.BUF_DEPTH, capi.BUF_NORMAL, capi.BUF_ROUGH_METAL, capi.BUF_VELOCITY, capi.BUF_RT_REFL, capi.BUF_RT_DIFF,
                        capi.BUF_FLT_RFL, capi.BUF_FLT_DFF, capi.BUF_TSS0, capi.BUF_TSS1, capi.BUF_BACKBUFFER):
                np.testing.assert_array_equal(a.context.readback(bid), b.context.readback(bid), err_msg="%dx%d buffer %d" % (size[0], size[1], bid))
            assert a.context.ray_count() == b.context.ray_count()
        finally:
            a.OnDestroy(); b.OnDestroy()


def _strips_through_rccl_equal_the_full_frame(W, H, world, balance, frames, mesh="bunny.obj", extra=(), peers=True, overreach=None):
    """`world` strips of one process, each its own context, exchanging through the direct RCCL path (raytracedggx_amd/rccl.py:
    ncclSend/ncclRecv in one group on the renderer's stream, pointers from StripRenderer.raw_ops) on the one GPU of the box: a
    single-rank communicator whose sends and receives pair up with each other -- against the single-context frame.  (Across
    processes the only difference is the peer number.)  peers: every strip maps every strip's history images (rtggx_set_history_peers,
    round 4), and the exchange carries the ordering tokens.  overreach: a list that receives, per frame, the largest number of rows by
    which a history tap of any strip read beyond the exchanged apron."""
    import torch
    from raytracedggx_amd import capi, rccl
    from raytracedggx_amd.strips import HISTORY_APRON, StripRenderer
    mesh, env = assets.path(mesh), assets.path("rnl_cross.dds")
    strips = []
    comm = rccl.Communicator(None, 0, 1)

    def transport(r, plan):
        ops = []
        for op, name, r0, r1, peer in plan:
            if op == "recv":
                src = strips[peer]
                ops += src.raw_ops([("send", name, r0, r1, 0)], src.context.frame_parity())
                ops += r.raw_ops([("recv", name, r0, r1, 0)], r.context.frame_parity())
        for t in strips:
            r.xstream.wait_stream(t.xstream); r.xstream.wait_stream(t.stream)
        comm.exchange(ops, r.xstream.cuda_stream)

    full = StripRenderer(W, H, mesh, env, extra_args=("-sharedmem",) + tuple(extra))
    strips += [StripRenderer(W, H, mesh, env, rank=r, world=world, transport=transport, torch_buffers=True, extra_args=("-sharedmem",) + tuple(extra), balance=balance, peers=peers) for r in range(world)]
    for t in strips:
        t.connect_peers(strips)
    if balance is True:
        assert all(s.bounds == strips[0].bounds for s in strips) and strips[0].bounds != [(r * H) // world for r in range(world + 1)]
        for _ in range(StripRenderer.PROFILE_FRAMES):          # the strips have rendered these as whole frames: the reference follows
            full.frame()
    try:
        for f in range(frames):
            full.frame()
            for s in strips:
                s.render()
            for s in strips:
                s.exchange()
            for s in strips:
                for t in strips:
                    s.stream.wait_stream(t.stream); s.stream.wait_stream(t.xstream)
            torch.cuda.synchronize(); full.context.sync()
            if overreach is not None:
                overreach.append(max(t.history_overreach(reset=True) for t in strips))
            np.testing.assert_array_equal(strips[0].context.readback(capi.BUF_BACKBUFFER), full.context.readback(capi.BUF_BACKBUFFER), err_msg="frame %d" % f)
            bid = capi.BUF_TSS1 if full.context.frame_parity() else capi.BUF_TSS0
            ref = full.context.readback(bid)
            for k, s in enumerate(strips):          # each strip's history, with the apron rows it received, equals the full frame's
                lo, hi = max(s.b - HISTORY_APRON, 0), min(s.e + HISTORY_APRON, H)
                assert s.context.frame_parity() == full.context.frame_parity()
                np.testing.assert_array_equal(s.context.readback(bid)[lo:hi], ref[lo:hi], err_msg="history of strip %d, frame %d" % (k, f))
        assert sum(s.context.ray_count() for s in strips) == full.context.ray_count(), "rays are counted once, by the strip that owns the pixel"
        return full.context.ray_count()
    finally:
        comm.destroy()
        full.close()
        for s in strips:
            s.close()


@pytest.mark.parametrize("world,balance", [(2, False), (8, False), (8, True), (5, [0, 40, 58, 120, 190, 272])],
                         ids=["2 strips", "8 strips", "8 balanced strips", "5 uneven strips"])
def test_strip_exchange_through_rccl_send_recv(built, world, balance):
    """With 8 strips the middle ones have two neighbours and strip 0 assembles seven others -- the shape of the 8-GPU run;
    `balanced`: every strip first profiles two whole frames and cuts the frame where the covered pixels balance, as bench.py
    does for N > 1."""
    _strips_through_rccl_equal_the_full_frame(480, 272, world, balance, 3)


def test_strips_equal_the_single_context_at_any_velocity(built):
    """SURVEY 8e's acceptance check -- "N-strip output == 1-strip output on every buffer" -- where rounds 2-3 failed it: 1280x720 with
    diffuse rays (-metallic 0.25 0.5), 8 balanced strips, 60 frames.  Now and then a sliver triangle at a silhouette gives one pixel a
    velocity of 40+ rows per frame (profiles/r03_l_soak.txt: the first difference was frame 37 of this very run); its history tap lands
    beyond the 18 exchanged rows, the apron guard counts it (rtggx_history_overreach > 0: observed here) and, since round 4, the tap
    reads the image of the strip that owns the row (rtggx_set_history_peers; the reference samples its one history texture anywhere,
    CSTemporalSS.hlsl:259-265): every frame's assembled back buffer and every strip's history stay bit-identical to the single context."""
    over = []
    _strips_through_rccl_equal_the_full_frame(1280, 720, 8, True, 60, extra=("-metallic", 0.25, 0.5), overreach=over)
    assert max(over) > 0, "no history tap went beyond the apron in 60 frames: the run no longer exercises what it is here for"
    print("frames with a history tap beyond the exchanged apron: %s" % [(f, n) for f, n in enumerate(over) if n])


def test_history_images_of_another_process_through_hip_ipc(built, tmp_path):
    """One process per GPU is how the strips run; a rank reads another rank's history image through a hipIpc mapping
    (rtggx_history_ipc_export / _open -> rtggx_set_history_peers).  Two PROCESSES on the one GPU of the box: the child renders whole
    frames; this process renders the upper half as a strip with NO apron exchanged at all (apron 0), so every history tap that crosses
    the boundary -- the bilinear footprint of the last row in every frame, any motion -- reads the child's image through the mapping.
    Frame by frame in lock step, its rows of TemporalSSOut and of the back buffer must be the child's."""
    import subprocess, sys
    from raytracedggx_amd import app, capi
    W, H, frames = 640, 360, 5
    args = ["-mesh", assets.path("bunny.obj"), "-env", assets.path("rnl_cross.dds"), "-width", W, "-height", H, "-sharedmem", "-dt", 0.1]
    child = subprocess.Popen([sys.executable, os.path.join(os.path.dirname(__file__), "ipc_peer_child.py"), str(tmp_path)] + [str(a) for a in args],
                             stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
    a = None
    try:
        line = child.stdout.readline().strip()
        assert line.startswith("handles "), line
        handles = bytes.fromhex(line.split()[1])
        assert len(handles) == 2 * capi.IPC_HANDLE_BYTES
        a = app.RayTracedGGX(args)
        ctx = a.context
        p0, p1 = ctx.history_ipc_open(handles)
        ctx.set_strip(0, H // 2); ctx.set_history_apron(0)
        ctx.set_history_peers([0, H // 2, H], [0, p0], [0, p1])
        for f in range(frames):
            child.stdin.write("frame\n"); child.stdin.flush()
            line = child.stdout.readline().strip()      # the child has rendered frame f and waited for it
            assert line == "done %d" % f, line
            a.OnUpdate(); a.OnRender(); ctx.sync()
            par = ctx.frame_parity()
            ref = np.load(tmp_path / ("tss_%d.npy" % f)); refbb = np.load(tmp_path / ("bb_%d.npy" % f))
            np.testing.assert_array_equal(ctx.readback(capi.BUF_TSS0 + par)[:H // 2], ref[:H // 2], err_msg="frame %d: history rows of the strip" % f)
            np.testing.assert_array_equal(ctx.readback(capi.BUF_BACKBUFFER)[:H // 2], refbb[:H // 2], err_msg="frame %d: back buffer rows of the strip" % f)
        assert ctx.history_overreach() >= 1, "taps across the boundary were counted (and read from the other process's image)"
        child.stdin.write("quit\n"); child.stdin.flush()
        assert child.wait(timeout=60) == 0
    finally:
        if child.poll() is None:
            child.kill()
        if a is not None:
            a.OnDestroy()


def test_c4_bunny_4k_full_frame_and_eight_strips(built):
    """BASELINE.json configs[3]: bunny at 3840x2160 (uffizi_cross.dds is not in the reference tree: rnl_cross.dds stands in),
    screen-tiled over 8 ranks.  Three full frames with the whole parity check against the oracle and a fourth with the property set
    (integer buffers, ray count, denoised image, determinism, strip independence), then the 8-rank shape on the one GPU of the box: 8 balanced strips exchanging through RCCL, every
    frame's assembled back buffer and every strip's history bit-identical to the single-context frame."""
    _full_size_properties(3840, 2160, "bunny.obj", "C4", checked_frames=3).close()      # (round 3 checked frame 0 only: no history in play)
    rays = _strips_through_rccl_equal_the_full_frame(3840, 2160, 8, True, 2)
    assert rays > 1500000


def test_c5_dragon_512_frames_and_4k_strips(built):
    """BASELINE.json configs[4]: 512 consecutive frames of the turning dragon at dt = 1/60 (the reference's only animation is the
    rigid rotation of RayTracer.cpp:270-272: 136 degrees over the run; FrameIndex wraps at 256, the Halton counter does not), the
    oracle carried along at 160x90 with the full parity check at frames 0, 255, 256 and 511; then the configuration's shape --
    3840x2160 over 8 strips -- for 6 frames against the single-context frame."""
    p = Pair(160, 90, mesh="dragon.obj", shared_mem=True)
    try:
        for f in range(512):
            p.frame()
            if f in (0, 255, 256, 511):
                p.check_frame("C5 frame %d" % f)
        assert p.app.frame_constants().view(np.uint32)[111] == 511 % 256      # CBGlobal::FrameIndex at byte 444
    finally:
        p.close()
    _strips_through_rccl_equal_the_full_frame(3840, 2160, 8, False, 6, mesh="dragon.obj")


def test_c5_deforming_dragon_4k_eight_strips(built):
    """BASELINE.json configs[4] as written: an ANIMATED dragon (-deform 0.3: new vertices and an asynchronous BVH refit every frame, on
    every rank), 3840x2160, 8 balanced strips exchanging through the RCCL group -- 4 frames, the assembled back buffer and every
    strip's history bit-identical to the single-context frame.  (Strip AND deforming: the tone map stays on the caller's stream, the
    traversals do not alternate streams -- the refit stream is busy.)"""
    rays = _strips_through_rccl_equal_the_full_frame(3840, 2160, 8, True, 4, mesh="dragon.obj", extra=("-deform", 0.3))
    assert rays > 1500000


def test_caller_owned_stream_with_two_traversals_in_flight(built):
    """A caller-owned main stream (rtggx_set_stream: what the strip exchange uses) at a size where launches are SMALL -- the traversals
    of odd frames go to a second stream, two in flight -- free-running for 40 frames against the same frames on the library's own
    stream, synchronised one by one: every target bit-identical.  Then the same with work of the caller's own ordered behind every frame
    on that stream (a copy of the back buffer: it must see the finished frame)."""
    import torch
    from raytracedggx_amd import app, capi
    args = ["-mesh", assets.path("bunny.obj"), "-env", assets.path("rnl_cross.dds"), "-width", 1920, "-height", 171, "-sharedmem", "-metallic", 1.0, 0.5]
    a, b = app.RayTracedGGX(args), app.RayTracedGGX(args)
    stream = torch.cuda.Stream()
    try:
        b.context.set_stream(stream.cuda_stream)
        bb = torch.as_tensor(_DeviceView(b.context.buffer_ptr(capi.BUF_BACKBUFFER), (171, 1920), "<u4"), device="cuda").view(torch.int32)
        copies = []
        for f in range(40):
            a.OnUpdate(); a.OnRender(); a.context.sync()
            if f >= 30: copies.append((a.context.readback(capi.BUF_BACKBUFFER), None))
            b.OnUpdate(); b.OnRender()
            if f >= 30:
                with torch.cuda.stream(stream):
                    copies[-1] = (copies[-1][0], bb.clone())
        b.context.sync(); torch.cuda.synchronize()
        for bid in (capi.BUF_VISIBILITY, capi.BUF_DEPTH, capi.BUF_NORMAL, capi.BUF_ROUGH_METAL, capi.BUF_VELOCITY, capi.BUF_RT_REFL, capi.BUF_RT_DIFF,
                    capi.BUF_FLT_RFL, capi.BUF_FLT_DFF, capi.BUF_TSS0, capi.BUF_TSS1, capi.BUF_BACKBUFFER):
            np.testing.assert_array_equal(a.context.readback(bid), b.context.readback(bid), err_msg="buffer %d" % bid)
        for k, (want, got) in enumerate(copies):
            np.testing.assert_array_equal(got.cpu().numpy().view(np.uint32), want, err_msg="the caller's copy behind frame %d" % (30 + k))
    finally:
        a.OnDestroy(); b.OnDestroy()


def test_long_run_past_the_frame_index_wrap(built):
    """C5-style run: 264 consecutive frames at dt = 1/60 (the model turns by 70 degrees; CBGlobal::FrameIndex wraps at
    256, the Halton counter does not, the constant slots and input sets cycle many times), the oracle carried along
    the whole way; full parity check at the start, around the wrap and at the end."""
    p = Pair(160, 90, metallic=(1.0, 0.5))
    try:
        for f in range(264):
            p.frame()
            if f in (0, 1, 2, 127, 254, 255, 256, 257, 263):
                p.check_frame("long run frame %d" % f)
        assert p.app.frame_constants().view(np.uint32)[111] == 263 % 256      # CBGlobal::FrameIndex at byte 444
    finally:
        p.close()


def test_deforming_mesh_by_reupload_and_rebuild(built):
    """SURVEY 8f rank 4, the functional part: a mesh that changes shape every frame (a travelling sine wave through the
    bunny's vertices) by re-uploading it and rebuilding its acceleration structure (rtggx_set_mesh + rtggx_build_as,
    both synchronous) -- every frame against the oracle given the same vertices and the same tree."""
    p = Pair(320, 180, metallic=(1.0, 0.5))
    try:
        v0, idx, _ = O.obj_import(assets.path("bunny.obj"))
        for f in range(4):
            v = v0.copy()
            if f:                                                    # frame 0: the mesh as loaded
                v[:, 0] += 0.35 * np.sin(1.3 * v0[:, 1] + 0.9 * f)
                v[:, 2] += 0.25 * np.cos(0.8 * v0[:, 1] - 0.7 * f)
                p.ctx.set_mesh(p.capi.MODEL_OBJ if hasattr(p.capi, "MODEL_OBJ") else 1, v, idx)
                p.ctx.build_as()
                p.o.set_mesh(1, v, idx)
                p.give_oracle_the_device_trees()                     # structure-checked, then handed over
            p.frame()
            p.check_frame("deforming frame %d" % f)
        vis0 = p.ctx.readback(p.capi.BUF_VISIBILITY)
        assert (vis0 >> 24 == 1).sum() > 1000                        # the (deformed) model is on screen
    finally:
        p.close()


def test_scripted_camera_and_material_track(built, tmp_path):
    """the sample's interactions as a script (-track file / RayTracedGGX::LoadTrack): an orbit with the
    left button held, a dolly, [DOWN] lowering the ground's metallic (diffuse rays appear), [V] switching the filter
    variant -- disocclusion and reprojection stress for the temporal pass, every frame against the oracle; [F11] in frame 3
    (RayTracedGGX.cpp:388-390, 703-717) writes that frame's back buffer as <prefix>_f000003.png in the middle of the run."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import imgdiff
    track = tmp_path / "orbit.track"
    track.write_text("# frame command args\n1 down 160 90\n1 move 150 86\n2 move 128 80\n2 key DOWN\n3 move 100 84\n3 wheel 2\n3 key F11\n4 up 0 0\n4 move 10 10\n4 key V\n5 key DOWN\n")
    p = Pair(320, 180)
    try:
        assert p.app.load_track(str(track))
        p.app.set_dump_prefix(str(tmp_path / "shot"))
        eyes = []
        for f in range(6):
            if f == 2: p.o.set_metallic(0, 0.75)                  # what [DOWN] does to mesh 0 in frame 2 ...
            if f == 5: p.o.set_metallic(0, 0.5)                   # ... and again in frame 5
            p.frame()
            p.check_frame("track frame %d" % f)
            eyes.append(p.app.frame_constants().tobytes())
            if f == 3:
                shot = p.app.last_screen_shot()
                assert shot == str(tmp_path / "shot_f000003.png") and os.path.exists(shot)
                bb = p.ctx.readback(p.capi.BUF_BACKBUFFER)
                np.testing.assert_array_equal(imgdiff.load(shot), np.stack([bb & 255, (bb >> 8) & 255, (bb >> 16) & 255], axis=-1).astype(np.uint8))
            else:
                assert (p.app.last_screen_shot() != "") == (f > 3)      # one event, one file
        assert len(set(eyes[1:4])) == 3 and eyes[4][:704] != eyes[3][:704]      # the camera moved while the button was held
        assert p.ctx.ray_count() > 1.5 * 320 * 180 * 0.2                        # diffuse rays are being traced by now
    finally:
        p.close()


def test_frame_dump_png_and_ppm_equal_the_back_buffer(built, tmp_path):
    """RayTracedGGX::SaveImage (the sample's screenshot, RayTracedGGX.cpp:719-739): PNG and PPM dumps hold the back buffer."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import imgdiff
    from raytracedggx_amd import app, capi
    a = app.RayTracedGGX(["-mesh", assets.path("bunny.obj"), "-env", assets.path("rnl_cross.dds"), "-width", 333, "-height", 217])
    try:
        for _ in range(2):
            a.OnUpdate(); a.OnRender()
        a.context.sync()
        bb = a.context.readback(capi.BUF_BACKBUFFER)
        want = np.stack([bb & 255, (bb >> 8) & 255, (bb >> 16) & 255], axis=-1).astype(np.uint8)
        assert want.std() > 10                        # a picture, not a constant
        for ext in ("png", "ppm"):
            path = str(tmp_path / ("shot." + ext))
            assert a.save_image(path)
            np.testing.assert_array_equal(imgdiff.load(path), want)
    finally:
        a.OnDestroy()


def test_c_abi_error_behaviour(built):
    """Misuse is reported through the return code + rtggx_last_error(), never by crashing or by silently doing nothing:
    calls out of order, bad arguments, short buffers (INTEGRATION.md "Error behaviour")."""
    import ctypes as C
    from raytracedggx_amd import capi
    L = capi.load()
    with pytest.raises(capi.RtggxError, match="bad arguments"):
        capi.Context(0, 16)
    ctx = capi.Context(64, 48)
    try:
        for call, what in ((ctx.render_visibility, "no frame constants"), (ctx.update_as, "rtggx_update_frame has not been called"),
                           (ctx.denoise, "no frame constants"), (ctx.tone_map, "no frame constants"), (ctx.timings, "timing not enabled")):
            with pytest.raises(capi.RtggxError, match=what):
                call()
        with pytest.raises(capi.RtggxError, match="bad rows"):
            ctx.set_strip(10, 49)
        with pytest.raises(capi.RtggxError, match="out of range"):
            ctx.set_mesh(1, np.zeros((3, 6), np.float32), np.array([0, 1, 3], np.uint32))
        with pytest.raises(capi.RtggxError, match="bad arguments"):
            ctx.set_mesh(2, np.zeros((3, 6), np.float32), np.array([0, 1, 2], np.uint32))
        ctx.update_frame(np.zeros(768, np.uint8))
        with pytest.raises(capi.RtggxError, match="rtggx_build_as has not been called"):
            ctx.ray_trace()
        small = np.zeros(16, np.uint32)
        assert L.rtggx_readback(ctx.h, capi.BUF_NORMAL, small.ctypes.data_as(C.c_void_p), small.nbytes) == -1
        assert b"needs" in L.rtggx_last_error()
        with pytest.raises(capi.RtggxError, match="not writable"):
            ctx.upload(capi.BUF_TLAS, np.zeros((2, 4, 4), np.float32))
        assert L.rtggx_sync(None) == -1 and b"null context" in L.rtggx_last_error()
        # and the context is still usable afterwards
        ctx.set_mesh(1, np.array([[-1, 0, 0, 0, 0, 1], [1, 0, 0, 0, 0, 1], [0, 2, 0, 0, 0, 1]], np.float32), np.array([0, 1, 2], np.uint32))
        ctx.build_as(); ctx.sync()
        assert ctx.bvh_root(1) == -1 and ctx.buffer_size(capi.BUF_BVH_TRIS1) == 64     # one triangle: the root is the leaf ~0
    finally:
        ctx.close()


def test_environment_upload_formats(built):
    """rtggx_set_env: float32 and float16 cube maps (DDS layout: per face its whole mip chain) land in the decoded
    RGBA16F environment (mip-major, 6 faces per mip) with round-to-nearest-even conversion; wrong sizes are refused."""
    from raytracedggx_amd import capi
    ctx = capi.Context(32, 32)
    try:
        rng = np.random.default_rng(5)
        size, mips = 4, 3
        faces = [[rng.uniform(0.0, 8.0, (max(size >> m, 1), max(size >> m, 1), 4)).astype(np.float32) for m in range(mips)] for _ in range(6)]
        dds32 = np.concatenate([faces[f][m].reshape(-1) for f in range(6) for m in range(mips)])
        want = np.concatenate([faces[f][m].astype(np.float16).reshape(-1, 4) for m in range(mips) for f in range(6)]).view(np.uint16)
        ctx.set_env(capi.FORMAT_RGBA32F, size, mips, dds32)
        np.testing.assert_array_equal(ctx.readback(capi.BUF_ENV), want)
        ctx.set_env(capi.FORMAT_RGBA16F, size, mips, dds32.astype(np.float16))
        np.testing.assert_array_equal(ctx.readback(capi.BUF_ENV), want)
        # BC6H, unsigned and signed: random blocks (every mode id, reserved ones included) against the oracle's decoder
        size, mips = 8, 2                                          # 2x2 blocks + 1 block per face
        blocks = rng.integers(0, 256, (6, 5, 16), dtype=np.uint8)   # per face: its mip chain, 4 + 1 blocks
        for fmt, signed in ((capi.FORMAT_BC6H_UF16, False), (capi.FORMAT_BC6H_SF16, True)):
            ctx.set_env(fmt, size, mips, blocks.reshape(-1))
            got = ctx.readback(capi.BUF_ENV).reshape(-1, 4)
            want = np.zeros((6 * 64 + 6 * 16, 4), np.uint16); want[:, 3] = 0x3C00
            for f in range(6):
                for b in range(4):
                    px = O.bc6h_decode_block(blocks[f, b].tobytes(), signed=signed).reshape(4, 4, 3)
                    face = want[f * 64:(f + 1) * 64].reshape(8, 8, 4)
                    face[(b // 2) * 4:(b // 2) * 4 + 4, (b % 2) * 4:(b % 2) * 4 + 4, :3] = px
                want[6 * 64 + f * 16:6 * 64 + (f + 1) * 16, :3] = O.bc6h_decode_block(blocks[f, 4].tobytes(), signed=signed)
            np.testing.assert_array_equal(got, want, err_msg="BC6H %s" % ("SF16" if signed else "UF16"))
        size, mips = 4, 3
        with pytest.raises(capi.RtggxError):
            ctx.set_env(capi.FORMAT_RGBA32F, size, mips, dds32[:-4])
        with pytest.raises(capi.RtggxError, match="unsupported format"):
            ctx.set_env(28, size, mips, dds32)           # DXGI_FORMAT_R8G8B8A8_UNORM
    finally:
        ctx.close()


def test_bc6h_sf16_dds_file_through_the_host_loader(built, tmp_path):
    """A BC6H_SF16 cube map file (the blocks of rnl_cross.dds with the DX10 header's format field set to 96: the same bits
    read as the signed variant) through the host's DDS reader and the GPU decoder, against the oracle's reader and
    decoder: every texel of all nine mips."""
    from raytracedggx_amd import app, capi
    raw = bytearray(open(assets.path("rnl_cross.dds"), "rb").read())
    assert int.from_bytes(raw[128:132], "little") == 95
    raw[128:132] = (96).to_bytes(4, "little")
    dds = tmp_path / "signed_cross.dds"
    dds.write_bytes(bytes(raw))
    a = app.RayTracedGGX(["-mesh", assets.path("triangle.obj"), "-env", str(dds), "-width", 64, "-height", 64])
    o = O.Oracle(64, 64)
    try:
        o.set_env_dds(str(dds))
        _, _, want = o.env_texels()
        got = a.context.readback(capi.BUF_ENV)
        np.testing.assert_array_equal(got, want)
        assert (got[:, :3] & 0x8000).any()               # negative halves do occur: it is the signed decode
        o2 = O.Oracle(64, 64); o2.set_env_dds(assets.path("rnl_cross.dds"))
        assert not np.array_equal(o2.env_texels()[2], want)
        o2.close()
    finally:
        a.OnDestroy(); o.close()


def test_context_lifecycle_and_mode_changes(built):
    """Contexts can be created and destroyed repeatedly, and the per-frame switches of the sample (the [V] shared-memory
    toggle, metallic changes) can flip between frames without disturbing parity."""
    import torch
    free0 = None
    for cycle in range(7):
        p = Pair(160, 96)
        p.frame()
        p.close()
        if cycle == 0:                      # the first context also loads the code objects and warms the runtime's pools
            torch.cuda.synchronize(); free0 = torch.cuda.mem_get_info()[0]
    torch.cuda.synchronize()
    assert free0 - torch.cuda.mem_get_info()[0] < 32 << 20, "device memory is returned on destroy"
    p = Pair(320, 180)
    try:
        for f, (shared, metal) in enumerate(((False, (1.0, 1.0)), (True, (0.25, 0.5)), (False, (0.25, 0.5)), (True, (1.0, 0.0)))):
            if shared != getattr(p, "_shared", False):
                p.app.OnKeyUp(ord("V"))                 # the sample's shared-memory toggle
            p._shared = shared
            for mesh, m in enumerate(metal):
                p.ctx.set_metallic(mesh, m); p.o.set_metallic(mesh, m)
            p.frame(); p.check_frame("mode change frame %d" % f)
    finally:
        p.close()


def test_timing_mode_free_running_equals_synchronised(built):
    """rtggx_enable_timing(1) uploads the frame constants in rtggx_update_as on stream B; an all-metal 1080p frame runs its
    visibility pass on stream C, which must be ordered behind that upload (it reads the constants).  40 free-running frames with
    per-pass timing on against the same frames synchronised one by one: every target bit-identical."""
    from raytracedggx_amd import app, capi
    args = ["-mesh", assets.path("bunny.obj"), "-env", assets.path("rnl_cross.dds"), "-width", 1920, "-height", 1080, "-sharedmem"]
    a, b = app.RayTracedGGX(args), app.RayTracedGGX(args)
    try:
        b.context.enable_timing(1)
        for f in range(40):
            a.OnUpdate(); a.OnRender(); a.context.sync()
            b.OnUpdate(); b.OnRender()
        b.context.sync()
        t = b.context.timings()
        assert 0.0 < t["visibility"] < 5.0 and 0.0 < t["ray_trace"] < 5.0
        for bid in (capi.BUF_VISIBILITY, capi.BUF_DEPTH, capi.BUF_NORMAL, capi.BUF_VELOCITY, capi.BUF_RT_REFL, capi.BUF_TSS0, capi.BUF_TSS1, capi.BUF_BACKBUFFER):
            np.testing.assert_array_equal(a.context.readback(bid), b.context.readback(bid), err_msg="buffer %d" % bid)
    finally:
        a.OnDestroy(); b.OnDestroy()


def test_small_frames_two_traversals_in_flight_equal_synchronised(built):
    """Launches below 200 000 rays send the traversals of odd frames to a second stream, so that two are in flight (capi.hip
    rtggx_ray_trace): 48 free-running frames of a turning 640x360 bunny (with diffuse rays: two rays per pixel, both image targets)
    and of a 1920x171 frame against the same frames synchronised one by one -- every target bit-identical, and the free-running
    ray total equal to the synchronised one."""
    from raytracedggx_amd import app, capi
    for size, extra in (((640, 360), ["-metallic", 0.25, 0.5]), ((1920, 171), [])):
        args = ["-mesh", assets.path("bunny.obj"), "-env", assets.path("rnl_cross.dds"), "-width", size[0], "-height", size[1], "-sharedmem", "-dt", 0.05] + extra
        a, b = app.RayTracedGGX(args), app.RayTracedGGX(args)
        try:
            for f in range(48):
                a.OnUpdate(); a.OnRender(); a.context.sync()
                b.OnUpdate(); b.OnRender()
            b.context.sync()
            assert a.context.ray_total() == b.context.ray_total() > 0
            for bid in (capi.BUF_VISIBILITY, capi.BUF_DEPTH, capi.BUF_NORMAL, capi.BUF_VELOCITY, capi.BUF_RT_REFL, capi.BUF_RT_DIFF, capi.BUF_FLT_RFL, capi.BUF_TSS0, capi.BUF_TSS1,
                        capi.BUF_BACKBUFFER):
                np.testing.assert_array_equal(a.context.readback(bid), b.context.readback(bid), err_msg="%dx%d buffer %d" % (size + (bid,)))
        finally:
            a.OnDestroy(); b.OnDestroy()


def test_diffuse_image_carry_hands_over_between_ray_generation_and_shading(built):
    """RayTracingOut1 keeps what it held where no diffuse ray is traced; with several input sets that is a carry-over from the previous set,
    done by ray generation while the previous frame's shading kernel wrote nothing into that set and by the shading kernel otherwise
    (capi.hip rtggx_ray_trace: genCarriesDiff / shadeWroteDiff).  Metal -> diffuse -> metal again, with the camera turning (pixels change
    from sky to covered and back): every frame against the oracle, which has ONE such image -- and the same schedule free-running against
    synchronised at a size that shades on the traversal's streams (640x360) and at one that shades on the main stream (1920x1080)."""
    from raytracedggx_amd import app, capi
    schedule = {3: 0.5, 5: 1.0, 9: 0.25, 10: 1.0}      # frame -> metallic of the ground from that frame on
    p = Pair(320, 180)
    try:
        for f in range(13):
            if f in schedule: p.ctx.set_metallic(0, schedule[f]); p.o.set_metallic(0, schedule[f])
            p.frame()
            p.check_frame("carry frame %d" % f)
    finally:
        p.close()
    for size in ((640, 360), (1920, 1080)):
        args = ["-mesh", assets.path("bunny.obj"), "-env", assets.path("rnl_cross.dds"), "-width", size[0], "-height", size[1], "-sharedmem", "-dt", 0.05]
        a, b = app.RayTracedGGX(args), app.RayTracedGGX(args)
        try:
            for f in range(16):
                if f in schedule: a.context.set_metallic(0, schedule[f]); b.context.set_metallic(0, schedule[f])
                a.OnUpdate(); a.OnRender(); a.context.sync()
                b.OnUpdate(); b.OnRender()
            b.context.sync()
            for bid in (capi.BUF_ROUGH_METAL, capi.BUF_RT_REFL, capi.BUF_RT_DIFF, capi.BUF_FLT_RFL, capi.BUF_FLT_DFF, capi.BUF_TSS0, capi.BUF_TSS1, capi.BUF_BACKBUFFER):
                np.testing.assert_array_equal(a.context.readback(bid), b.context.readback(bid), err_msg="%dx%d buffer %d" % (size + (bid,)))
        finally:
            a.OnDestroy(); b.OnDestroy()


def test_trace_workgroup_size_changes_nothing(built):
    """The traversal's resident workgroup has 12 waves, or 14 / 16 when a trial says so (trace.hip steerTraceWaves); launches with few
    rays use single-wave workgroups.  rtggx_debug_trace_residency pins the size: five frames of the turning dragon with diffuse rays at
    every size against the default -- every ray-dependent target bit-identical, the ray totals equal; an unsupported size is an error."""
    from raytracedggx_amd import app, capi
    args = ["-mesh", assets.path("dragon.obj"), "-env", assets.path("rnl_cross.dds"), "-width", 1280, "-height", 720, "-sharedmem", "-dt", 0.1, "-metallic", 0.25, 0.5]
    ref = app.RayTracedGGX(args)
    try:
        for f in range(5):
            ref.OnUpdate(); ref.OnRender()
        ref.context.sync()
        want = {bid: ref.context.readback(bid) for bid in (capi.BUF_RT_REFL, capi.BUF_RT_DIFF, capi.BUF_FLT_DFF, capi.BUF_TSS0, capi.BUF_TSS1, capi.BUF_BACKBUFFER)}
        total = ref.context.ray_total()
        assert ref.context.trace_residency()[0] in (12, 14, 16)
        with pytest.raises(capi.RtggxError):
            ref.context.trace_residency(11)
    finally:
        ref.OnDestroy()
    for waves in (10, 12, 14, 16):
        a = app.RayTracedGGX(args)
        try:
            assert a.context.trace_residency(waves)[0] == waves
            for f in range(5):
                a.OnUpdate(); a.OnRender()
            a.context.sync()
            assert a.context.trace_residency()[0] == waves, "a pinned size stays"
            assert a.context.ray_total() == total
            for bid, w in want.items():
                np.testing.assert_array_equal(a.context.readback(bid), w, err_msg="%d waves, buffer %d" % (waves, bid))
        finally:
            a.OnDestroy()


@pytest.mark.parametrize("W,H,rows,extra", [(1920, 1080, None, ()), (333, 201, None, ("-metallic", 0.5, 0.5)), (640, 360, (100, 231), ()), (640, 360, (0, 14), ()), (640, 360, (346, 360), ())],
                         ids=["1080p", "ragged 333x201 with diffuse rays", "strip of rows 100-231", "strip of the first 14 rows", "strip of the last 14 rows"])
def test_fused_temporal_tone_map_equals_the_two_kernels(built, W, H, rows, extra):
    """Round 4: the temporal pass can tone-map its result as well (denoise.hip temporalToneKernel; rtggx_tone_map then finds its work done;
    the library does so on small launches).  Twelve free-running frames with rtggx_debug_fuse_tone_map(ctx, 1) against twelve with
    (ctx, 0) -- temporalKernel + toneMapKernel, rounds 1-3's path: the back buffer, both history images (the strip's apron rows included) and the filtered image bit-identical; and a tone map
    that did NOT follow a denoise in its frame, or follows an upload, still runs as a kernel of its own."""
    from raytracedggx_amd import app, capi
    args = ["-mesh", assets.path("bunny.obj"), "-env", assets.path("rnl_cross.dds"), "-width", W, "-height", H, "-sharedmem"] + list(extra)
    a, b = app.RayTracedGGX(args), app.RayTracedGGX(args)
    try:
        a.context.fuse_tone_map(True); b.context.fuse_tone_map(False)
        for c in (a.context, b.context):
            if rows:
                c.set_strip(*rows)
        for f in range(12):
            a.OnUpdate(); a.OnRender()
            b.OnUpdate(); b.OnRender()
        a.context.sync(); b.context.sync()
        for bid in (capi.BUF_BACKBUFFER, capi.BUF_TSS0, capi.BUF_TSS1, capi.BUF_FLT_DFF):
            np.testing.assert_array_equal(a.context.readback(bid), b.context.readback(bid), err_msg="buffer %d" % bid)
        if rows is None:
            # a tone map of uploaded data: no denoise in front of it, a kernel of its own
            c = a.context
            tss = c.readback(capi.BUF_TSS0 + c.frame_parity())
            want = c.readback(capi.BUF_BACKBUFFER)
            c.upload(capi.BUF_BACKBUFFER, np.zeros_like(want))
            c.tone_map(); c.sync()
            np.testing.assert_array_equal(c.readback(capi.BUF_BACKBUFFER), want, err_msg="a second tone map of the same frame")
            c.upload(capi.BUF_TSS0 + c.frame_parity(), np.zeros_like(tss))
            c.tone_map(); c.sync()
            assert not c.readback(capi.BUF_BACKBUFFER)[2:-2, 2:-2].any(), "the tone map of an uploaded (black) image is black"
    finally:
        a.OnDestroy(); b.OnDestroy()


def test_copy_bandwidth_leaves_the_context_as_it_found_it(built):
    """rtggx_copy_bandwidth (the bench line's measured peak) in the middle of a free-running run changes nothing."""
    from raytracedggx_amd import app, capi
    args = ["-mesh", assets.path("bunny.obj"), "-env", assets.path("rnl_cross.dds"), "-width", 1280, "-height", 720, "-sharedmem"]
    a, b = app.RayTracedGGX(args), app.RayTracedGGX(args)
    try:
        for f in range(10):
            if f == 5:
                gbs = b.context.copy_bandwidth(1 << 28, 2)
                assert 500.0 < gbs < 8000.0, gbs      # read + written bytes per second of a device-to-device copy: below the 8 TB/s of the HBM
            a.OnUpdate(); a.OnRender(); a.context.sync()
            b.OnUpdate(); b.OnRender()
        b.context.sync()
        for bid in (capi.BUF_BACKBUFFER, capi.BUF_TSS0, capi.BUF_TSS1, capi.BUF_FLT_DFF, capi.BUF_RT_REFL):
            np.testing.assert_array_equal(a.context.readback(bid), b.context.readback(bid), err_msg="buffer %d" % bid)
    finally:
        a.OnDestroy(); b.OnDestroy()


def test_every_stream_placement_free_running_equals_synchronised(built):
    """Where a frame's kernels go is decided in one place from five facts (capi.hip placeFrame: small launch, strip, deforming mesh,
    diffuse rays, caller-owned main stream).  Every one of the 32 keys at 320x180: the placement is the table's, and twelve free-running
    frames equal twelve frames synchronised one by one in every target.  (`small` is pinned with rtggx_debug_placement: at this size
    the ray count alone would always say small.)"""
    import itertools
    import torch
    from raytracedggx_amd import app, capi
    W, H = 320, 180
    v0, idx, _ = O.obj_import(assets.path("bunny.obj"))
    stream = torch.cuda.Stream()
    targets = (capi.BUF_VISIBILITY, capi.BUF_DEPTH, capi.BUF_NORMAL, capi.BUF_ROUGH_METAL, capi.BUF_VELOCITY, capi.BUF_RT_REFL, capi.BUF_RT_DIFF,
               capi.BUF_FLT_DFF, capi.BUF_TSS0, capi.BUF_TSS1, capi.BUF_BACKBUFFER)
    for small, strip, deforming, diffuse, caller in itertools.product((False, True), repeat=5):
        args = ["-mesh", assets.path("bunny.obj"), "-env", assets.path("rnl_cross.dds"), "-width", W, "-height", H, "-sharedmem"] + (["-metallic", 1.0, 0.5] if diffuse else [])
        a, b = app.RayTracedGGX(args), app.RayTracedGGX(args)
        label = "small %d strip %d deforming %d diffuse %d caller-owned stream %d" % (small, strip, deforming, diffuse, caller)
        try:
            for c in (a.context, b.context):
                c.placement(1 if small else 0)
                if strip:
                    c.set_strip(40, 140)
            if caller:
                b.context.set_stream(stream.cuda_stream)
            for f in range(12):
                for x, sync in ((a, True), (b, False)):
                    if deforming:
                        x.context.refit_as(1, _wave(v0, f))
                    x.OnUpdate(); x.OnRender()
                    if sync:
                        x.context.sync()
            b.context.sync(); torch.cuda.synchronize()
            key, where = b.context.placement(1 if small else 0)
            assert key == {"small": small, "strip": strip, "deforming": deforming, "diffuse": diffuse, "caller_stream": caller}, label
            want = {"raster": "C", "gen": "C", "trace": "B" if (not small or deforming) else "R",      # the twelfth frame is frame 12: even -> B ... (below)
                    "shade": "main", "frames_in_flight": 3 if (not small and (deforming or diffuse)) else 4}
            if small and not deforming:
                want["trace"] = "B"          # frame counter 12 is even: stream B; the odd frames before it went to R
                want["shade"] = "B"          # small launches shade on the traversal's stream
            assert where == want, "%s: %s, the table says %s" % (label, where, want)
            for bid in targets:
                np.testing.assert_array_equal(a.context.readback(bid), b.context.readback(bid), err_msg="%s, buffer %d" % (label, bid))
        finally:
            a.OnDestroy(); b.OnDestroy()


def test_tile_words_follow_strips_uploads_and_skipped_passes(built):
    """Round 4: the visibility pass keeps a word per 16x16 tile of its target ("something was drawn here") and the kernels behind it leave
    tiles whose word is 0 after a scalar load (rtggx_debug_tile_words; rtggx_context.h visDirtyBuf).  The words are only as good as their
    bookkeeping: tiles are counted from the pass's first row, the target is cleared two frames ahead by another frame's ray generation, a
    caller may upload a visibility buffer or skip a pass.  One context with the words, one without, through the same schedule of strip
    changes (rows that are no multiple of 16, growing and shrinking), a frame without ray tracing and an uploaded visibility buffer, and a model that changes shape -- every target of every frame identical inside the strip."""
    from raytracedggx_amd import app, capi
    W, H = 320, 180
    args = ["-mesh", assets.path("bunny.obj"), "-env", assets.path("rnl_cross.dds"), "-width", W, "-height", H, "-sharedmem", "-metallic", 1.0, 0.5, "-dt", 0.25]
    a, b = app.RayTracedGGX(args), app.RayTracedGGX(args)
    targets = ((capi.BUF_VISIBILITY, 1), (capi.BUF_DEPTH, 1), (capi.BUF_NORMAL, 1), (capi.BUF_ROUGH_METAL, 1), (capi.BUF_VELOCITY, 1), (capi.BUF_RT_REFL, 1), (capi.BUF_RT_DIFF, 1),
               (capi.BUF_FLT_DFF, 1), (capi.BUF_TSS0, 1), (capi.BUF_TSS1, 1), (capi.BUF_BACKBUFFER, 1))
    schedule = [(0, H)] * 3 + [(40, 140)] * 3 + [(37, 150)] * 3 + [(37, 120)] * 2 + [(0, H)] * 3 + [(8, H)] * 2 + [(8, 100)] * 3 + [(0, H)] * 2 + [(0, H)] * 6 + [(21, 163)] * 3
    v0, _, _ = O.obj_import(assets.path("bunny.obj"))
    try:
        b.context.tile_words(False)
        rows = None
        for f, r in enumerate(schedule):
            for x in (a, b):
                c = x.context
                if r != rows:
                    c.set_strip(*r)
                if f >= 21:                                           # ... and the model changes shape from here on (a refit per frame, per input set)
                    c.refit_as(1, _wave(v0, f))
                x.OnUpdate()
                if f == 7:                                            # a frame whose passes the caller issues one by one
                    c.render_visibility(); c.update_as(); c.ray_trace(); c.denoise(True); c.tone_map()
                elif f == 12:                                         # a visibility pass and nothing behind it
                    c.render_visibility()
                elif f == 16:                                         # the caller's own visibility buffer (the one just rendered, handed back)
                    c.render_visibility(); c.sync()
                    c.upload(capi.BUF_VISIBILITY, c.readback(capi.BUF_VISIBILITY))
                    c.update_as(); c.ray_trace(); c.denoise(True); c.tone_map()
                else:
                    x.OnRender()
            rows = r
            if f % 3 == 2 or f in (7, 12, 16):
                a.context.sync(); b.context.sync()
                for bid, _ in targets:
                    ia, ib = a.context.readback(bid), b.context.readback(bid)
                    np.testing.assert_array_equal(ia.reshape(H, W)[r[0]:r[1]], ib.reshape(H, W)[r[0]:r[1]], err_msg="frame %d rows %s buffer %d" % (f, r, bid))
    finally:
        a.OnDestroy(); b.OnDestroy()


def test_update_as_after_render_visibility(built):
    """The C ABI allows rtggx_update_as after rtggx_render_visibility of the same frame (the sample overlaps the two on its two
    queues, RayTracedGGX.cpp:304-339): the visibility pass has then carried the slot to the device with the previous frame's TLAS,
    and rtggx_ray_trace must send the refreshed one.  Six frames of the turning bunny issued in that order against the usual order."""
    from raytracedggx_amd import app, capi
    args = ["-mesh", assets.path("bunny.obj"), "-env", assets.path("rnl_cross.dds"), "-width", 640, "-height", 360, "-sharedmem", "-dt", 0.25]   # 4 degrees per frame
    a, b = app.RayTracedGGX(args), app.RayTracedGGX(args)
    try:
        for f in range(6):
            a.OnUpdate(); a.OnRender(); a.context.sync()
            b.OnUpdate()                                             # UpdateFrame -> rtggx_update_frame
            c = b.context
            c.render_visibility(); c.update_as(); c.ray_trace(); c.denoise(True); c.tone_map(); c.sync()
            np.testing.assert_array_equal(c.readback(capi.BUF_TLAS), a.context.readback(capi.BUF_TLAS))
            for bid in (capi.BUF_VISIBILITY, capi.BUF_NORMAL, capi.BUF_RT_REFL, capi.BUF_TSS0, capi.BUF_TSS1, capi.BUF_BACKBUFFER):
                np.testing.assert_array_equal(a.context.readback(bid), c.readback(bid), err_msg="frame %d buffer %d" % (f, bid))
        assert not np.array_equal(a.context.readback(capi.BUF_TLAS)[1], np.eye(4, dtype=np.float32))
    finally:
        a.OnDestroy(); b.OnDestroy()


def test_sync_flag_and_async_toggle_change_nothing(built):
    """`-sync` / key [A] (m_asyncCompute, RayTracedGGX.cpp:394-396): the frame on ONE stream in submission order against the
    multi-stream frame, toggled mid-run as well -- every target bit-identical."""
    from raytracedggx_amd import app, capi
    base = ["-mesh", assets.path("bunny.obj"), "-env", assets.path("rnl_cross.dds"), "-width", 1280, "-height", 720, "-sharedmem", "-metallic", 1.0, 0.5]
    a, b, c = app.RayTracedGGX(base), app.RayTracedGGX(base + ["-sync"]), app.RayTracedGGX(base)
    try:
        for f in range(12):
            if f in (4, 9):
                c.OnKeyUp(ord("A"))
            for x in (a, b, c):
                x.OnUpdate(); x.OnRender()
        for x in (a, b, c):
            x.context.sync()
        for bid in (capi.BUF_VISIBILITY, capi.BUF_DEPTH, capi.BUF_NORMAL, capi.BUF_ROUGH_METAL, capi.BUF_VELOCITY, capi.BUF_RT_REFL, capi.BUF_RT_DIFF,
                    capi.BUF_FLT_RFL, capi.BUF_FLT_DFF, capi.BUF_TSS0, capi.BUF_TSS1, capi.BUF_BACKBUFFER):
            ref = a.context.readback(bid)
            np.testing.assert_array_equal(b.context.readback(bid), ref, err_msg="-sync, buffer %d" % bid)
            np.testing.assert_array_equal(c.context.readback(bid), ref, err_msg="[A] toggled, buffer %d" % bid)
    finally:
        a.OnDestroy(); b.OnDestroy(); c.OnDestroy()


def test_headless_executable_with_the_bat_file_arguments(built, tmp_path):
    """The RayTracedGGX executable (host/Main.cpp) with the argument tail of Bin/Bunny.bat -- `-mesh Assets/bunny.obj 0.0 0.0 0.0 1.0`,
    relative to a Bin/-like working directory, default environment `Assets/rnl_cross.dds` -- plus `-frames 3 -dump`: its PNG
    equals the back buffer of the same three frames driven through the library."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import imgdiff
    from raytracedggx_amd import app, capi
    exe = os.path.join(os.path.dirname(os.path.abspath(app.HOST_LIB_PATH)), "RayTracedGGX")
    bin_dir = assets.asset_dir()                                      # holds Assets/, like the reference's Bin/
    bat_tail = "-mesh Assets/bunny.obj 0.0 0.0 0.0 1.0".split()       # Bin/Bunny.bat:1
    shot = tmp_path / "exe_frame.png"
    r = subprocess.run([exe] + bat_tail + ["-width", "640", "-height", "360", "-frames", "3", "-dump", str(shot)], cwd=bin_dir, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "3 frames 640x360" in r.stdout and shot.exists()
    a = app.RayTracedGGX(["-mesh", assets.path("bunny.obj"), "0.0", "0.0", "0.0", "1.0", "-env", assets.path("rnl_cross.dds"), "-width", 640, "-height", 360])
    try:
        for _ in range(3):
            a.OnUpdate(); a.OnRender()
        a.context.sync()
        bb = a.context.readback(capi.BUF_BACKBUFFER)
        want = np.stack([bb & 255, (bb >> 8) & 255, (bb >> 16) & 255], axis=-1).astype(np.uint8)
        np.testing.assert_array_equal(imgdiff.load(str(shot)), want)
    finally:
        a.OnDestroy()
    # and a refused flag says so instead of rendering something else
    r = subprocess.run([exe, "-mesh", "Assets/missing.obj"], cwd=bin_dir, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "cannot import" in r.stderr


def test_executable_strips_in_one_process_equal_the_single_frame(built, tmp_path):
    """The C++ multi-GPU host (host/Strips.cpp; `-gpus N` restarts the executable once per GPU) in its single-process mode:
    `-strips N` renders N strips with N contexts on the one GPU of the box and carries out every rank's exchange plan through real
    ncclSend / ncclRecv (a one-rank communicator: across processes only the peer numbers differ).  The frame rank 0 assembles --
    equal strips, and strips balanced by two profile frames -- is the single-context frame, byte for byte; the plan functions
    agree with raytracedggx_amd/strips.py."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import imgdiff
    from raytracedggx_amd import app
    exe = os.path.join(os.path.dirname(os.path.abspath(app.HOST_LIB_PATH)), "RayTracedGGX")
    bin_dir = assets.asset_dir()
    common = ["-mesh", "Assets/bunny.obj", "-width", "640", "-height", "360", "-sharedmem", "-dt", "0.05"]
    def run(extra, name):
        shot = tmp_path / name
        r = subprocess.run([exe] + common + extra + ["-dump", str(shot)], cwd=bin_dir, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        return imgdiff.load(str(shot)), r.stdout
    single5, _ = run(["-frames", "5"], "single5.png")
    equal4, out = run(["-frames", "5", "-strips", "4", "-balance", "0"], "equal4.png")
    assert "4 strips in one process" in out and "boundaries 0 90 180 270 360" in out
    np.testing.assert_array_equal(equal4, single5)
    single7, _ = run(["-frames", "7"], "single7.png")
    balanced6, out = run(["-frames", "5", "-strips", "6"], "balanced6.png")      # + the two profile frames every strip renders whole
    np.testing.assert_array_equal(balanced6, single7)
    bounds = [int(x) for x in out.split("boundaries")[1].split(":")[0].split()]
    assert len(bounds) == 7 and bounds[0] == 0 and bounds[-1] == 360 and bounds != [(r * 360) // 6 for r in range(7)]
    # the same boundaries as the Python host cuts from the same profile
    from raytracedggx_amd import capi
    from raytracedggx_amd.strips import StripRenderer
    s = StripRenderer(640, 360, assets.path("bunny.obj"), assets.path("rnl_cross.dds"), rank=0, world=6, transport=lambda *_: None, extra_args=("-sharedmem", "-dt", 0.05), balance=True)
    try:
        assert s.bounds == bounds
    finally:
        s.close()


def test_history_apron_guard_reports_fast_motion(built, tmp_path):
    """SURVEY 8e "clamp and report": strips exchange HISTORY_APRON (18) rows of last frame's temporal result, enough for 16 px of
    vertical reprojection per frame.  An orbit drag (-track) moves the image faster: the temporal pass of a strip then reads
    rows it was never given -- it must SAY so (rtggx_history_overreach), and an apron widened by the reported amount makes the
    two strips bit-identical to the single-context frame again; so does, at the default apron, reading such taps from the owner's image
    (rtggx_set_history_peers, round 4)."""
    from raytracedggx_amd import capi
    from raytracedggx_amd.strips import HISTORY_APRON, StripRenderer
    W, H = 640, 360
    mesh, env = assets.path("bunny.obj"), assets.path("rnl_cross.dds")
    track = tmp_path / "drag.track"
    track.write_text("1 down 320 180\n2 move 320 168\n3 move 320 150\n4 move 320 138\n5 up 0 0\n")      # 12-18 px of vertical drag per frame: a pitch of 0.2-0.3 rad
    extra = ("-sharedmem", "-track", str(track))

    def run(apron, connect=False):
        strips = []

        def transport(r, plan):
            for op, name, r0, r1, peer in plan:
                if op == "recv":
                    bid = capi.BUF_BACKBUFFER if name == "backbuffer" else capi.BUF_TSS0 + r.context.frame_parity()
                    mine = r.context.readback(bid)
                    mine[r0:r1] = strips[peer].context.readback(bid)[r0:r1]
                    r.context.upload(bid, mine)

        full = StripRenderer(W, H, mesh, env, extra_args=extra)
        strips += [StripRenderer(W, H, mesh, env, rank=r, world=2, transport=transport, extra_args=extra, apron=apron) for r in range(2)]
        if connect:
            for s in strips:
                s.connect_peers(strips)
        try:
            equal = True
            for f in range(6):
                full.frame(); full.context.sync()
                for s in strips:
                    s.render(); s.context.sync()
                for s in strips:
                    s.exchange()
                equal = equal and np.array_equal(strips[0].context.readback(capi.BUF_BACKBUFFER), full.context.readback(capi.BUF_BACKBUFFER))
            return max(s.history_overreach() for s in strips), equal
        finally:
            full.close()
            for s in strips:
                s.close()

    over, equal = run(HISTORY_APRON)
    assert over > 0, "the drag is faster than the default apron covers, and the guard says by how much"
    over2, equal2 = run(HISTORY_APRON + over + 1)
    assert over2 == 0 and equal2, "with the apron widened by the reported amount the strips are exact again (%d rows over, equal: %s)" % (over2, equal2)
    over3, equal3 = run(HISTORY_APRON, connect=True)
    assert over3 == over and equal3, "round 4: with the strips' history images mapped into each other the same taps are still counted, and harmless"


def _wave(v0, f, amp=0.35):
    v = v0.copy()
    v[:, 0] += amp * np.sin(1.3 * v0[:, 1] + 0.9 * f)
    v[:, 2] += 0.7 * amp * np.cos(0.8 * v0[:, 1] - 0.7 * f)
    return v


def test_deforming_mesh_async_refit_against_the_oracle(built):
    """SURVEY 8f rank 4 / BASELINE config 5's "async BVH refit": rtggx_refit_as stages new vertices; the upload, the new leaf
    triangles and the bottom-up box refit of the EXISTING tree run on the refit stream at the start of the next frame.  Every frame
    against the oracle given the same vertices and the refitted tree (structure-checked: boxes tight around the moved
    triangles), including frames WITHOUT a new shape in between (the other input sets' vertex buffers must follow) and a
    change violent enough to make the library REBUILD the tree -- beside the frames, a few launches per frame on the refit stream
    (round 3): the frames while it is in progress and the frames after the new topology has taken over are checked like the others."""
    p = Pair(320, 180, metallic=(1.0, 0.5), shared_mem=True)
    try:
        p.ctx.set_refit_policy(1.6, 16)      # (the mild wave below must stay under the threshold, the violent change must exceed it)
        v0, idx, _ = O.obj_import(assets.path("bunny.obj"))
        shapes = {1: _wave(v0, 1), 2: _wave(v0, 2), 3: _wave(v0, 3), 6: _wave(v0, 6), 7: _wave(v0, 7)}      # frames 4, 5: no new shape
        big = v0.copy(); big[:, 0] *= 3.0; big[:, 1] = v0[:, 1] * (1.0 + 0.6 * np.sin(3.0 * v0[:, 0]))     # frame 8: a very different shape

        def frame(f, shape):
            if shape is not None:
                p.ctx.refit_as(1, shape)
                p.o.set_mesh(1, shape, idx)
            p.app.OnUpdate(); p.app.OnRender(); p.ctx.sync()
            p.give_oracle_the_device_trees(refitted=True)            # the refitted (or rebuilt) tree of this frame's input set, structure-checked
            p.o.set_frame_constants(p.app.frame_constants().tobytes()[:704] + p.o.get_frame_constants().tobytes()[704:])
            p.o.update_as(); p.o.render_visibility(); p.rays = p.o.ray_trace(); p.o.denoise(); p.o.tone_map()
            p.check_frame("refit frame %d" % f)
            return p.ctx.refit_stats(1)

        for f in range(8):
            st = frame(f, shapes.get(f))
        # 5 new shapes + frames 4 and 5, whose input sets still held an older shape: vertices copied over and refitted as well
        assert st["refits"] == 7 and st["rebuilds"] == 0 and 0.8 < st["cost_ratio"] < 1.6, st
        f, after = 8, 0
        while after < 5 and f < 60:          # the rebuild takes a handful of frames; then five more on the new topology (every input set gets its tree)
            st = frame(f, big if f == 8 else _wave(big, f, 0.2))
            after += 1 if st["rebuilds"] >= 1 else 0
            f += 1
        assert st["rebuilds"] >= 1, "the violent change made the refitted tree's cost drift past the threshold: %s" % st
        assert st["cost_ratio"] < 1.4, "the rebuilt topology fits the new shape: %s" % st
        with pytest.raises(p.capi.RtggxError, match="vertices given"):
            p.ctx.refit_as(1, v0[:-1])
    finally:
        p.close()


def test_build_as_while_a_mesh_deforms(built):
    """rtggx_build_as on a context whose model deforms (replacing the other mesh: rtggx_set_mesh(ground) + rtggx_build_as, or simply
    building again): the deforming mesh keeps its per-set vertex buffers and trees -- built from its newest shape, every input set's
    tree from that set's own vertices --, and later refits keep writing per-set trees.  Every frame against the oracle; then the
    same sequence free-running against synchronised, bit-identical."""
    from raytracedggx_amd import app, capi
    v0, idx, _ = O.obj_import(assets.path("bunny.obj"))
    slab_v = np.array([[-1, 1, -1, 0, 1, 0], [1, 1, -1, 0, 1, 0], [1, 1, 1, 0, 1, 0], [-1, 1, 1, 0, 1, 0]], np.float32) * np.array([1, 1, 1, 1, 1, 1], np.float32)
    slab_i = np.array([3, 1, 0, 2, 1, 3], np.uint32)      # a one-sided ground: two triangles instead of the slab's twelve
    p = Pair(320, 180, metallic=(1.0, 0.5), shared_mem=True)
    try:
        for f in range(9):
            shape = _wave(v0, f) if f else None
            if shape is not None:
                p.ctx.refit_as(1, shape); p.o.set_mesh(1, shape, idx)
            if f == 4:       # replace the ground and build: the model is deforming at this point
                p.ctx.set_mesh(0, slab_v, slab_i); p.ctx.build_as()
                p.o.set_mesh(0, slab_v, slab_i); p.num_tris[0] = 2
            if f == 6:
                p.ctx.build_as()
            p.app.OnUpdate(); p.app.OnRender(); p.ctx.sync()
            p.give_oracle_the_device_trees(refitted=True)
            p.o.set_frame_constants(p.app.frame_constants().tobytes()[:704] + p.o.get_frame_constants().tobytes()[704:])
            p.o.update_as(); p.o.render_visibility(); p.rays = p.o.ray_trace(); p.o.denoise(); p.o.tone_map()
            p.check_frame("build_as while deforming, frame %d" % f)
    finally:
        p.close()
    args = ["-mesh", assets.path("bunny.obj"), "-env", assets.path("rnl_cross.dds"), "-width", 640, "-height", 360, "-sharedmem", "-metallic", 1.0, 0.5]
    a, b = app.RayTracedGGX(args), app.RayTracedGGX(args)
    try:
        for f in range(14):
            for x, sync in ((a, True), (b, False)):
                x.context.refit_as(1, _wave(v0, f, 0.3))
                if f == 5:
                    x.context.set_mesh(0, slab_v, slab_i); x.context.build_as()
                if f == 9:
                    x.context.build_as()
                x.OnUpdate(); x.OnRender()
                if sync:
                    x.context.sync()
        b.context.sync()
        for bid in (capi.BUF_VISIBILITY, capi.BUF_NORMAL, capi.BUF_RT_REFL, capi.BUF_RT_DIFF, capi.BUF_FLT_DFF, capi.BUF_TSS0, capi.BUF_TSS1, capi.BUF_BACKBUFFER):
            np.testing.assert_array_equal(a.context.readback(bid), b.context.readback(bid), err_msg="free-running, buffer %d" % bid)
    finally:
        a.OnDestroy(); b.OnDestroy()


def test_deforming_mesh_free_running_equals_synchronised(built):
    """The refit needs no synchronisation: 30 frames of a breathing bunny issued back to back (three frames in flight, per-set
    vertex buffers, staging ring) against the same frames synchronised one by one -- every target bit-identical; and the
    `-deform` flag of the host (RayTracedGGX::OnUpdate -> RayTracer::UpdateMesh) against explicit rtggx_refit_as calls."""
    from raytracedggx_amd import app, capi
    args = ["-mesh", assets.path("bunny.obj"), "-env", assets.path("rnl_cross.dds"), "-width", 1280, "-height", 720, "-sharedmem", "-metallic", 1.0, 0.5]
    a, b, c = app.RayTracedGGX(args), app.RayTracedGGX(args), app.RayTracedGGX(args + ["-deform", 0.3])
    try:
        v0, idx, _ = app.obj_import(assets.path("bunny.obj"))
        period = 32
        shapes = []
        for k in range(period):           # what host/RayTracedGGX.cpp computes for -deform 0.3 (fp32, same formula)
            phase = np.float32(6.283185307) * np.float32(k) / np.float32(period)
            v = v0.copy()
            v[:, 0] = v0[:, 0] + np.float32(0.3) * np.sin(np.float32(1.3) * v0[:, 1] + phase, dtype=np.float32)
            v[:, 2] = v0[:, 2] + np.float32(0.7) * np.float32(0.3) * np.cos(np.float32(0.8) * v0[:, 1] - phase, dtype=np.float32)
            shapes.append(v)
        for x in (a, b, c):
            x.context.set_refit_policy(4.0, 16)      # no rebuild in this test: when a rebuild's topology takes over depends on timing, and the trees are compared below
        for f in range(30):
            a.context.refit_as(1, shapes[f % period]); a.OnUpdate(); a.OnRender(); a.context.sync()
            b.context.refit_as(1, shapes[f % period]); b.OnUpdate(); b.OnRender()
            c.OnUpdate(); c.OnRender()
        b.context.sync(); c.context.sync()
        for bid in (capi.BUF_VISIBILITY, capi.BUF_DEPTH, capi.BUF_NORMAL, capi.BUF_ROUGH_METAL, capi.BUF_VELOCITY, capi.BUF_RT_REFL, capi.BUF_RT_DIFF,
                    capi.BUF_FLT_RFL, capi.BUF_FLT_DFF, capi.BUF_TSS0, capi.BUF_TSS1, capi.BUF_BACKBUFFER, capi.BUF_BVH4_NODES1, capi.BUF_BVH_TRIS1):
            np.testing.assert_array_equal(a.context.readback(bid), b.context.readback(bid), err_msg="free-running, buffer %d" % bid)
        assert a.context.ray_count() == b.context.ray_count()
        st = b.context.refit_stats(1)
        assert st["refits"] == 30 and st["rebuilds"] == 0, st
        # the host's own animation: sin/cos of libm in fp32 may differ from numpy's in the last bit, so compare the geometry-independent
        # counters and the image loosely, and the integer visibility coverage closely
        assert c.context.refit_stats(1)["refits"] == 30
        va, vc = a.context.readback(capi.BUF_VISIBILITY), c.context.readback(capi.BUF_VISIBILITY)
        assert (va != vc).mean() < 2e-3, "-deform renders the same animation (%.4f%% of the visibility words differ)" % (100 * (va != vc).mean())
    finally:
        a.OnDestroy(); b.OnDestroy(); c.OnDestroy()


def test_deforming_mesh_from_device_memory(built):
    """rtggx_refit_as_device (round 4): the vertices of a mesh animated ON the GPU, handed over as a device pointer with the stream that
    produces them.  A torch stream writes each frame's shape into ONE buffer, hands it over, and overwrites the buffer with garbage right
    behind the call -- as the next animation step would: 24 free-running frames against 24 synchronised frames that got the same shapes
    from host memory (rtggx_refit_as), every target and the model's tree bit-identical.  (The call orders its copy out of the buffer like
    a hipMemcpyAsync on the producer's stream: behind what the stream held, in front of what it is given next.)"""
    import torch
    from raytracedggx_amd import app, capi
    args = ["-mesh", assets.path("bunny.obj"), "-env", assets.path("rnl_cross.dds"), "-width", 640, "-height", 360, "-sharedmem", "-metallic", 1.0, 0.5]
    a, b = app.RayTracedGGX(args), app.RayTracedGGX(args)
    try:
        v0, idx, _ = app.obj_import(assets.path("bunny.obj"))
        shapes = [np.ascontiguousarray(_wave(v0, f), np.float32) for f in range(8)]
        on_gpu = [torch.from_numpy(s).cuda() for s in shapes]
        producer = torch.cuda.Stream()
        buf = torch.empty_like(on_gpu[0])
        for x in (a, b):
            x.context.set_refit_policy(4.0, 16)      # (no rebuild: when its topology takes over depends on timing)
        for f in range(24):
            a.context.refit_as(1, shapes[f % 8]); a.OnUpdate(); a.OnRender(); a.context.sync()
            with torch.cuda.stream(producer):
                buf.copy_(on_gpu[f % 8], non_blocking=True)
                b.context.refit_as_device(1, buf.data_ptr(), buf.shape[0], producer.cuda_stream)
                buf.fill_(float("nan"))           # the buffer is the producer's again at once
            b.OnUpdate(); b.OnRender()
        b.context.sync(); torch.cuda.synchronize()
        for bid in (capi.BUF_VISIBILITY, capi.BUF_DEPTH, capi.BUF_NORMAL, capi.BUF_VELOCITY, capi.BUF_RT_REFL, capi.BUF_RT_DIFF,
                    capi.BUF_FLT_DFF, capi.BUF_TSS0, capi.BUF_TSS1, capi.BUF_BACKBUFFER, capi.BUF_BVH4_NODES1, capi.BUF_BVH_TRIS1):
            np.testing.assert_array_equal(a.context.readback(bid), b.context.readback(bid), err_msg="buffer %d" % bid)
        assert b.context.refit_stats(1)["refits"] == 24
        with pytest.raises(capi.RtggxError):
            b.context.refit_as_device(1, buf.data_ptr(), buf.shape[0] - 1)
    finally:
        a.OnDestroy(); b.OnDestroy()
