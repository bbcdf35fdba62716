"""Pins the CPU oracle (oracle/) against every known answer available for this path.

The reference has no tests and no golden vectors of its own; what exists is
  * the integer known-answer values derived from the cited shader lines,
  * the ObjLoader::Import facts of SURVEY.md 8c (tests/golden/obj_import.json),
  * the published BC6H bit layout (hand-assembled blocks below),
  * identities the formulas must satisfy.
"""
import json
import os

import numpy as np
import pytest

import assets
from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def lib(built):
    return O.lib()


def test_pcg_known_answers(lib):
    # RayTracing.hlsl:379-387, SURVEY.md Appendix F
    kat = {0: 0x108EF29B, 1: 0x00033628, 2: 0xA3FE8633, 255: 0xD4CA7FD8, 0xFFFFFFFF: 0x106EE0AB, 2073599: 0x9C361D45}
    for seed, want in kat.items():
        assert lib.orc_rng(seed) == want


def test_sample_param_table(lib):
    # RayTracing.hlsl:394-406, SURVEY.md Appendix F: (x, y, W, FrameIndex) -> (s, xi.x, pcg(s)&0xFFFF)
    import ctypes as C
    table = [(0, 0, 1920, 0, 185, 0.72265625, 9455), (1, 0, 1920, 0, 224, 0.875, 64962), (0, 1, 1920, 0, 184, 0.71875, 26177),
             (959, 539, 1920, 0, 131, 0.51171875, 62012), (959, 539, 1920, 1, 30, 0.1171875, 28161),
             (1919, 1079, 1920, 255, 254, 0.9921875, 61415), (255, 255, 256, 7, 48, 0.1875, 59238), (3839, 2159, 3840, 128, 125, 0.48828125, 23647)]
    for x, y, W, fi, s, xix, lo16 in table:
        sv = C.c_uint32()
        xi = np.zeros(2, np.float32)
        lib.orc_sample_param(C.c_uint32(x), C.c_uint32(y), C.c_uint32(W), C.c_uint32(fi), C.byref(sv), xi.ctypes.data_as(C.c_void_p))
        assert sv.value == s and xi[0] == np.float32(xix) and xi[1] == np.float32(lo16 / 65536.0)


def test_incremental_halton(built):
    # SURVEY.md row H4 / Appendix F (fp32, incremental accumulation)
    o = O.Oracle(8, 8, threads=1)
    want_x = [0.5, 0.25, 0.75, 0.125, 0.625, 0.375, 0.875, 0.0625]
    want_y = ["0.33333334", "0.6666667", "0.11111112", "0.44444448", "0.7777778", "0.22222222", "0.5555556", "0.88888896"]
    for wx, wy in zip(want_x, want_y):
        x, y = o.halton()
        assert x == np.float32(wx)
        assert y == np.float32(wy), (y, wy)


def test_obj_import_golden_facts(built):
    # XUSGObjLoader.cpp:18-40, 72-431 -- facts recorded from the reference's own importer
    facts = json.load(open(os.path.join(HERE, "golden", "obj_import.json")))
    for name in ("bunny", "dragon", "TuringBowl"):
        f = facts[name]
        v, i, aabb = O.obj_import(assets.path(name + ".obj"))
        assert v.shape == (f["num_verts"], 6) and i.size == f["num_indices"]
        assert list(i[:6]) == f["first_indices"]
        np.testing.assert_allclose(aabb[:3], f["aabb_min"], atol=1e-4)
        np.testing.assert_allclose(aabb[3:], f["aabb_max"], atol=1e-4)
        assert "%08x" % assets.fnv1a32(i.tobytes()) == f["fnv_indices"]
        assert "%08x" % assets.fnv1a32(v.tobytes()) == f["fnv_verts"]
    v, _, _ = O.obj_import(assets.path("bunny.obj"))
    np.testing.assert_allclose(v[0, :3], facts["bunny"]["v0"], atol=1e-6)
    np.testing.assert_allclose(v[0, 3:], facts["bunny"]["n0"], atol=1e-6)


def test_obj_import_with_normals_negative_indices_and_fans(built, tmp_path):
    # v//vn faces (vertex splitting, XUSGObjLoader.cpp:300-335), negative indices (:243), polygon fan (:267-297)
    p = tmp_path / "quad.obj"
    p.write_text("# comment\nmtllib x.mtl\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nvn 0 0 2\n"
                 "f 1//1 2//1 3//1 4//1\nf -4//2 -3//2 -2//2\n")
    v, i, aabb = O.obj_import(str(p))
    # fan: (1,2,3),(1,3,4) then (1,2,3) with the second normal -> three split vertices; whole index array reversed
    assert i.size == 9 and v.shape[0] == 7
    assert list(i[::-1][:6]) == [0, 1, 2, 0, 2, 3]
    np.testing.assert_array_equal(v[:4, 5], [-1, -1, -1, -1])          # vn.z negated, normalised
    np.testing.assert_array_equal(v[4:, 5], [-1, -1, -1])              # split copies carry the second normal (0,0,2)/2, z negated
    np.testing.assert_array_equal(aabb, [0, 0, 0, 1, 1, 0])


def test_visibility_word_examples():
    # PSVisibility.hlsl:23, SURVEY.md Appendix F
    word = lambda inst, prim: ((inst << 24) | prim) + 1
    assert word(0, 0) == 0x00000001 and word(0, 11) == 0x0000000C and word(1, 0) == 0x01000001 and word(1, 69665) == 0x01011022


def test_format_conversions(lib):
    import ctypes as C
    # binary16: exhaustive round trip, and agreement with numpy's IEEE conversion on random floats
    for h in range(0, 0x7C00, 7):
        f = lib.orc_f16_to_f32(C.c_uint16(h))
        assert lib.orc_f32_to_f16(C.c_float(f)) == h
        assert f == float(np.array([h], np.uint16).view(np.float16)[0])
    rng = np.random.default_rng(1)
    xs = (rng.standard_normal(4000) * 10.0 ** rng.uniform(-9, 5, 4000)).astype(np.float32)
    with np.errstate(over="ignore"):
        want = xs.astype(np.float16).view(np.uint16)
    got = np.array([lib.orc_f32_to_f16(C.c_float(float(x))) for x in xs], np.uint16)
    np.testing.assert_array_equal(got, want)
    # R11G11B10_FLOAT: every code round-trips; negatives clamp to 0; overflow saturates to the largest finite value
    rgb = np.zeros(3, np.float32)
    for code in list(range(0, 0x7C0, 3)) + [0x7BF]:
        lib.orc_unpack_r11g11b10f(C.c_uint32(code | (code << 11) | ((code >> 1) << 22)), rgb.ctypes.data_as(C.c_void_p))
        assert lib.orc_pack_r11g11b10f(rgb.ctypes.data_as(C.c_void_p)) == code | (code << 11) | ((code >> 1) << 22)
    pk = lambda r, g, b: lib.orc_pack_r11g11b10f(np.array([r, g, b], np.float32).ctypes.data_as(C.c_void_p))
    assert pk(-1.0, -0.0, -1e30) == 0
    assert pk(1.0, 1.0, 1.0) == (15 << 6) | ((15 << 6) << 11) | ((15 << 5) << 22)
    assert pk(1e9, 65024.0, 64512.0) == 0x7BF | (0x7BF << 11) | (0x3DF << 22)
    assert pk(1.0 + 1.0 / 128, 1.0 + 3.0 / 128, 0.0) == ((15 << 6) | 0) | (((15 << 6) | 2) << 11)      # ties to even
    # R10G10B10A2_UNORM: floor(x*1023+0.5)
    p = lib.orc_pack_r10g10b10a2(np.array([0.5, 1.0, 0.25, 1.0], np.float32).ctypes.data_as(C.c_void_p))
    assert p == 512 | (1023 << 10) | (256 << 20) | (3 << 30)


def _bc6h_block(fields_bits):
    """Assemble a 128-bit block from (value, nbits) pairs, LSB first."""
    acc, pos = 0, 0
    for v, n in fields_bits:
        acc |= (v & ((1 << n) - 1)) << pos
        pos += n
    assert pos <= 128
    return acc.to_bytes(16, "little")


def test_bc6h_hand_assembled_blocks(built):
    # Mode 11 (0x03): 10-bit endpoints, one region, no transform, 4-bit indices (3 bits for texel 0).
    def unq(c):
        return 0 if c == 0 else (0xFFFF if c == 1023 else ((c << 16) + 0x8000) >> 10)
    w3 = [0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64]
    e0, e1 = (100, 200, 300), (700, 1023, 0)
    idx = [3, 0, 15, 7, 1, 2, 4, 5, 6, 8, 9, 10, 11, 12, 13, 14]
    fields = [(0x03, 5)] + [(c, 10) for c in e0] + [(c, 10) for c in e1] + [(idx[0], 3)] + [(i, 4) for i in idx[1:]]
    out = O.bc6h_decode_block(_bc6h_block(fields))
    for t in range(16):
        for c in range(3):
            a, b = unq(e0[c]), unq(e1[c])
            v = (a * (64 - w3[idx[t]]) + b * w3[idx[t]] + 32) >> 6
            assert out[t, c] == (v * 31) >> 6
    # Mode 12 (0x07): 11-bit base, 9-bit signed deltas; rw[10] sits after rx[8:0].
    base, delta = (0x4D2, 0x155, 0x7FF), (-5, 17, -256)
    fields = [(0x07, 5)] + [(b & 0x3FF, 10) for b in base]
    for b, d in zip(base, delta):
        fields += [(d & 0x1FF, 9), (b >> 10, 1)]
    fields += [(0, 3)] + [(15, 4)] * 15
    out = O.bc6h_decode_block(_bc6h_block(fields))
    def unq11(c):
        return 0 if c == 0 else (0xFFFF if c == 2047 else ((c << 16) + 0x8000) >> 11)
    for c in range(3):
        assert out[0, c] == (unq11(base[c]) * 31) >> 6                      # index 0 -> first endpoint
        assert out[5, c] == (unq11((base[c] + delta[c]) & 0x7FF) * 31) >> 6  # index 15 -> second endpoint


def test_bc6h_signed_hand_assembled_blocks(built):
    """BC6H_SF16: endpoints are two's-complement, magnitudes unquantise to 15 bits, the half carries the
    sign -- restated here from the format description, independently of the decoder, for the untransformed mode 11 and
    the transformed mode 12."""
    def sx(v, bits):
        v &= (1 << bits) - 1
        return v - (1 << bits) if v >> (bits - 1) else v
    def unq(c, bits):
        neg, c = c < 0, abs(c)
        u = 0 if c == 0 else (0x7FFF if c >= (1 << (bits - 1)) - 1 else ((c << 15) + 0x4000) >> (bits - 1))
        return -u if neg else u
    def fin(v):
        return (0x8000 | (((-v) * 31) >> 5)) if v < 0 else (v * 31) >> 5
    w4 = [0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64]
    e0, e1 = (-100, 200, -512), (511, -511, 0)          # 10-bit two's complement, extremes included
    idx = [3, 0, 15, 7, 1, 2, 4, 5, 6, 8, 9, 10, 11, 12, 13, 14]
    fields = [(0x03, 5)] + [(c, 10) for c in e0] + [(c, 10) for c in e1] + [(idx[0], 3)] + [(i, 4) for i in idx[1:]]
    out = O.bc6h_decode_block(_bc6h_block(fields), signed=True)
    for t in range(16):
        for c in range(3):
            a, b = unq(e0[c], 10), unq(e1[c], 10)
            v = (a * (64 - w4[idx[t]]) + b * w4[idx[t]] + 32) >> 6
            assert out[t, c] == fin(v), (t, c)
    assert out[1, 0] == fin(unq(-100, 10)) and out[1, 0] & 0x8000        # index 0: the first endpoint, negative
    # Mode 12 (0x07): 11-bit base (two's complement), 9-bit signed deltas, the sum wraps in 11 bits
    base, delta = (-700, 0x155, 1023), (-5, 17, 255)
    fields = [(0x07, 5)] + [(b & 0x3FF, 10) for b in base]
    for b, d in zip(base, delta):
        fields += [(d & 0x1FF, 9), ((b >> 10) & 1, 1)]
    fields += [(0, 3)] + [(15, 4)] * 15
    out = O.bc6h_decode_block(_bc6h_block(fields), signed=True)
    for c in range(3):
        assert out[0, c] == fin(unq(sx(base[c], 11), 11))
        assert out[5, c] == fin(unq(sx(base[c] + delta[c], 11), 11))      # 1023 + 255 wraps to a negative endpoint
    # the same block read as unsigned differs: the flag matters
    assert not np.array_equal(out, O.bc6h_decode_block(_bc6h_block(fields)))


def test_bc6h_mip_chain_is_consistent(built):
    # every used mode decodes sensibly: each mip is close to the 2x2 box filter of the previous one
    o = O.Oracle(8, 8, threads=1)
    o.set_env_dds(assets.path("rnl_cross.dds"))
    size, mips, tex = o.env_texels()
    assert (size, mips) == (256, 9) and tex.shape[0] == 6 * sum((256 >> m) ** 2 for m in range(9))
    f = tex.view(np.float16).astype(np.float32)
    off, prev = 0, None
    for m in range(mips):
        s = size >> m
        lv = f[off:off + 6 * s * s].reshape(6, s, s, 4)[..., :3]
        off += 6 * s * s
        assert np.isfinite(lv).all() and (lv >= 0).all()
        if prev is not None:
            ds = prev.reshape(6, s, 2, s, 2, 3).mean(axis=(2, 4))
            assert np.abs(ds - lv).mean() / np.abs(lv).mean() < 0.15
        prev = lv


def test_matrix_inverse_and_camera(built):
    import ctypes as C
    vp = O.camera_view_proj(1920, 1080)
    inv = np.zeros((4, 4), np.float32)
    O.lib().orc_matrix_inverse(vp.ctypes.data_as(C.c_void_p), inv.ctypes.data_as(C.c_void_p))
    np.testing.assert_allclose(vp.astype(np.float64) @ inv.astype(np.float64), np.eye(4), atol=2e-4)
    # the focus point (0,3,0) projects to the screen centre; near plane 1, far plane 1000 (RayTracedGGX.cpp:19-23, 267-277)
    p = np.array([0.0, 3.0, 0.0, 1.0]) @ vp.astype(np.float64)
    assert abs(p[0] / p[3]) < 1e-6 and abs(p[1] / p[3]) < 1e-6
    dist = np.linalg.norm(np.array([10.0, 10.0, -24.0]) - np.array([0.0, 3.0, 0.0]))
    assert abs(p[3] - dist) < 1e-4


def _fullscreen_setup(W, H, tris):
    """Oracle whose model mesh is `tris` given directly in NDC (identity WVP, zero jitter)."""
    o = O.Oracle(W, H, threads=1)
    verts = np.zeros((len(tris) * 3, 6), np.float32)
    verts[:, :3] = np.asarray(tris, np.float32).reshape(-1, 3)
    verts[:, 5] = -1
    o.set_mesh(1, verts, np.arange(len(tris) * 3, dtype=np.uint32))
    o.set_mesh(0, verts[:3] * 0, np.array([0, 0, 0], np.uint32))   # degenerate ground: culled
    fc = np.zeros(768, np.uint8)
    f = fc.view(np.float32)
    ident = np.eye(4, dtype=np.float32).reshape(-1)
    for base in (0, 16, 32, 48):                  # WorldViewProjs / Prev
        f[base:base + 16] = ident
    f[136:152] = ident                            # perObject[0].WorldViewProj  (byte 544)
    f[156:172] = ident                            # perObject[1].WorldViewProj  (byte 624)
    o.set_frame_constants(fc.tobytes())
    return o


def test_rasteriser_fill_rules(built):
    W = H = 16
    # NDC -> pixels: x_px = (x+1)*8, y_px = (1-y)*8.  A pixel-aligned quad [4,12)x[4,12) as two clockwise triangles
    q = lambda x, y: ((x / 8.0) - 1.0, 1.0 - (y / 8.0), 0.5)
    tris = [[q(4, 4), q(12, 4), q(12, 12)], [q(4, 4), q(12, 12), q(4, 12)]]
    o = _fullscreen_setup(W, H, tris)
    o.render_visibility()
    vis = o.buffer(O.BUF_VISIBILITY)
    cov = vis > 0
    assert cov.sum() == 64 and cov[4:12, 4:12].all()                         # top-left rule: right and bottom edges excluded
    assert set(np.unique(vis[cov])) == {0x01000001, 0x01000002}              # each pixel owned by exactly one triangle
    assert (o.buffer(O.BUF_DEPTH)[cov] == int(0.5 * 16777215.0 + 0.5)).all()
    assert (o.buffer(O.BUF_DEPTH)[~cov] == 0xFFFFFF).all()
    # counter-clockwise = back face: culled
    o = _fullscreen_setup(W, H, [[q(4, 4), q(12, 12), q(12, 4)]])
    o.render_visibility()
    assert not (o.buffer(O.BUF_VISIBILITY) > 0).any()
    # depth LESS: the nearer triangle wins wherever both cover; equal depth keeps the lower primitive id
    near = [(x, y, 0.25) for x, y, _ in tris[0]]
    o = _fullscreen_setup(W, H, [tris[0], near, tris[0]])
    o.render_visibility()
    vis = o.buffer(O.BUF_VISIBILITY)
    assert set(np.unique(vis[vis > 0])) == {0x01000002}
    o = _fullscreen_setup(W, H, [tris[0], tris[0]])
    o.render_visibility()
    assert set(np.unique(o.buffer(O.BUF_VISIBILITY)[o.buffer(O.BUF_VISIBILITY) > 0])) == {0x01000001}
    # depth clip: z outside [0,1] is discarded
    o = _fullscreen_setup(W, H, [[(x, y, 1.5) for x, y, _ in tris[0]]])
    o.render_visibility()
    assert not (o.buffer(O.BUF_VISIBILITY) > 0).any()


def test_rasteriser_guard_band(built):
    """A triangle with vertices 3 x 10^6 viewport half-widths away (snapped coordinates far beyond 2^30 sub-pixel units) is clipped against the
    guard band |x|, |y| <= 256 w (oracle/orc_raster.h holds the contract) and still covers every pixel, with the depth of its plane; one that
    lies wholly beyond one guard plane covers nothing."""
    W, H = 32, 16
    big = [(-3.0e6, 3.0e6, 0.25), (3.0e6, 3.0e6, 0.5), (0.0, -6.0e6, 0.75)]      # clockwise on the y-down screen = front
    o = _fullscreen_setup(W, H, [big])
    o.render_visibility()
    vis, depth = o.buffer(O.BUF_VISIBILITY).reshape(H, W), o.buffer(O.BUF_DEPTH).reshape(H, W)
    assert (vis == 0x01000001).all()
    # the plane z(x, y) through the three vertices, at the pixel centres
    A = np.array([[x, y, 1.0] for x, y, _ in big], np.float64)
    a, b, c = np.linalg.solve(A, np.array([z for _, _, z in big], np.float64))
    xs = (np.arange(W) + 0.5) / (W / 2.0) - 1.0
    ys = 1.0 - (np.arange(H) + 0.5) / (H / 2.0)
    want = a * xs[None, :] + b * ys[:, None] + c
    assert np.abs(depth.astype(np.float64) - want * 16777215.0).max() <= 4.0, np.abs(depth.astype(np.float64) - want * 16777215.0).max()
    far_right = [(400.0, 1.0, 0.5), (500.0, 1.0, 0.5), (450.0, -1.0, 0.5)]
    o = _fullscreen_setup(W, H, [far_right])
    o.render_visibility()
    assert not (o.buffer(O.BUF_VISIBILITY) > 0).any()
    # ... and one that straddles the right guard plane but not the viewport changes nothing inside it
    beside = [(2.0, 1.0, 0.5), (5.0e5, 1.0, 0.5), (3.0, -1.0, 0.5)]
    o = _fullscreen_setup(W, H, [beside])
    o.render_visibility()
    assert not (o.buffer(O.BUF_VISIBILITY) > 0).any()


def test_bvh_trace_equals_brute_force(built):
    # closest-hit semantics must not depend on the hierarchy (oracle/orc_bvh.h)
    o = O.Oracle(64, 64)
    v, i, _ = O.obj_import(assets.path("bunny.obj"))
    o.set_mesh(1, v, i)
    o.build_as()
    o.update_frame((10, 10, -24), O.camera_view_proj(64, 64), 0.25)
    o.update_as()
    rng = np.random.default_rng(7)
    n = 600
    org = np.array([10.0, 10.0, -24.0]) + rng.standard_normal((n, 3)) * 0.5
    tgt = np.stack([rng.uniform(-6, 6, n), rng.uniform(-1, 10, n), rng.uniform(-6, 6, n)], 1)
    rays = np.concatenate([org, tgt - org, np.full((n, 1), 1e-5), np.full((n, 1), 1e4)], 1).astype(np.float32)
    a, b = o.trace_rays(rays), o.trace_rays(rays, brute=True)
    assert a["valid"].sum() > n // 3
    np.testing.assert_array_equal(a["valid"], b["valid"])
    hit = a["valid"]
    np.testing.assert_array_equal(a["inst"][hit], b["inst"][hit])
    np.testing.assert_array_equal(a["prim"][hit], b["prim"][hit])
    np.testing.assert_array_equal(a["t"][hit], b["t"][hit])
    # degenerate interval [0,0] never hits (RayTracing.hlsl:429, 447-452)
    rays[:, 6:] = 0
    assert not o.trace_rays(rays)["valid"].any()


def test_config_c1_single_triangle_constant_env(built):
    """BASELINE.json configs[0]: single triangle + constant environment, 256x256, 1 frame, CPU only."""
    W = H = 256
    o = O.Oracle(W, H)
    v, i, _ = O.obj_import(assets.path("triangle.obj"))
    assert v.shape == (3, 6) and list(i) == [2, 1, 0]
    o.set_mesh(1, v, i)
    o.set_env_rgba16f(1, 1, assets.constant_env_rgba16f(1.0))
    o.build_as()
    o.transform_sh()
    sh = o.buffer(O.BUF_SH_COEFFS)
    np.testing.assert_allclose(sh[0], 2.0 * np.sqrt(np.pi), rtol=1e-5)     # constant radiance 1: L00 = sqrt(4*pi), others 0
    assert np.abs(sh[1:]).max() < 1e-5
    rays = o.render((10, 10, -24), O.camera_view_proj(W, H), 0.0)
    vis = o.buffer(O.BUF_VISIBILITY)
    assert (vis == 0).any() and ((vis > 0) & (vis < 0x01000000)).any() and (vis == 0x01000001).any()
    assert rays > 0
    refl = O.unpack_r11g11b10f(o.buffer(O.BUF_RT_REFL))
    assert (refl[vis == 0] == 1.0).all()                                    # background = environment(-V) = 1
    assert np.isfinite(refl).all() and (refl >= 0).all()
    bb = O.unpack_rgba8(o.buffer(O.BUF_BACKBUFFER))
    # background: tss = ITM(TM(1)) = 1 -> tone map 1/(1+0.5) with zero laplacian away from edges -> round(255*2/3) = 170
    inner = np.zeros_like(vis, bool); inner[2:-2, 2:-2] = True
    far_bg = inner.copy()   # 5x5 erosion: the temporal 3x3 window and the tone-map cross reach two pixels
    for dy in range(-2, 3):
        for dx in range(-2, 3):
            far_bg &= np.roll(np.roll(vis == 0, dy, 0), dx, 1)
    assert (np.abs(bb[far_bg][:, :3].astype(int) - 170) <= 1).all()
    # G-buffer normal of the ground's top face is +Y: (0.5, 1, 0.5) in R10G10B10A2
    top = (vis > 0) & (vis <= 2)
    assert (o.buffer(O.BUF_NORMAL)[top] == (512 | (1023 << 10) | (512 << 20) | (3 << 30))).all()
