"""CPU-side tests of the product: the C++ host layer (no GPU needed for the host arithmetic) and the
C ABI surface.  No compute entry point is called here."""
import ctypes as C
import json
import os
import re
import subprocess

import numpy as np
import pytest

import assets
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_every_declared_symbol(built):
    """librtggx.so loads and exports exactly what include/rtggx.h declares."""
    from raytracedggx_amd import capi
    header = open(os.path.join(ROOT, "include", "rtggx.h")).read()
    declared = sorted(set(re.findall(r"\b(rtggx_[a-z_0-9]+)\s*\(", header)))
    assert len(declared) >= 25
    lib = C.CDLL(capi.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), "librtggx.so does not export " + name
    assert sorted(capi.EXPORTS) == declared
    # the constant-buffer mirror has the reference's byte layout
    txt = subprocess.run(["nm", "-D", "--defined-only", capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (rtggx_[a-z_0-9]+)", txt))
    assert exported == set(declared)


def test_host_library_exports(built):
    from raytracedggx_amd import app
    lib = C.CDLL(app.HOST_LIB_PATH)
    for name in app.HOST_EXPORTS:
        assert hasattr(lib, name)


def test_product_fails_loudly_without_gpu(built):
    """No CPU fallback: creating a context without a HIP device is an error, not a silent downgrade."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from raytracedggx_amd import app, capi
    with pytest.raises(capi.RtggxError):
        capi.Context(64, 64)
    with pytest.raises(capi.RtggxError):
        app.RayTracedGGX(["-mesh", assets.path("triangle.obj"), "-env", assets.path("rnl_cross.dds"), "-width", 64, "-height", 64])


def test_package_asks_for_eight_hardware_queues_before_the_runtime_starts():
    """The frame runs on four HIP streams, RCCL and the strip exchange bring more; with HIP's default of four hardware queues a fifth
    stream shares one (slowest of eight strips 0.137 ms instead of 0.077; profiles/r03_h_strip_projection.txt).  Importing the package sets
    GPU_MAX_HW_QUEUES=8 unless the caller has chosen a value -- in a fresh interpreter, where nothing has initialised the runtime yet."""
    import subprocess, sys
    code = "import os; os.environ.pop('GPU_MAX_HW_QUEUES', None); import raytracedggx_amd; print(os.environ['GPU_MAX_HW_QUEUES'])"
    assert subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, check=True).stdout.strip() == "8"
    code = "import os; os.environ['GPU_MAX_HW_QUEUES'] = '6'; import raytracedggx_amd; print(os.environ['GPU_MAX_HW_QUEUES'])"
    assert subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, check=True).stdout.strip() == "6"


def test_host_obj_importer_matches_golden_and_oracle(built):
    from raytracedggx_amd import app
    facts = json.load(open(os.path.join(ROOT, "tests", "golden", "obj_import.json")))
    for name in ("bunny", "dragon", "TuringBowl"):
        v, i, aabb = app.obj_import(assets.path(name + ".obj"))
        assert "%08x" % assets.fnv1a32(i.tobytes()) == facts[name]["fnv_indices"]
        assert "%08x" % assets.fnv1a32(v.tobytes()) == facts[name]["fnv_verts"]
        ov, oi, oaabb = O.obj_import(assets.path(name + ".obj"))
        np.testing.assert_array_equal(v.view(np.uint32), ov.view(np.uint32))
        np.testing.assert_array_equal(i, oi)
        np.testing.assert_array_equal(aabb, oaabb)


def test_host_obj_importer_edge_cases(built, tmp_path):
    from raytracedggx_amd import app
    cases = {
        "normals_fan_negative.obj": "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nvn 0 0 2\nf 1//1 2//1 3//1 4//1\nf -4//2 -3//2 -2//2\n",
        "texcoords.obj": "v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 0 1\nf 1/1 2/2 3/3\n",
        "full.obj": "v 0 0 0\nv 2 0 0\nv 0 2 0\nv 0 0 2\nvt 0 0\nvn 0 0 1\nvn 1 0 0\nf 1/1/1 2/1/1 3/1/1\nf 1/1/2 3/1/2 4/1/2\n# end\n",
        "empty.obj": "# nothing here\n",
    }
    for name, text in cases.items():
        p = tmp_path / name
        p.write_text(text)
        v, i, aabb = app.obj_import(str(p))
        ov, oi, oaabb = O.obj_import(str(p))
        np.testing.assert_array_equal(v.view(np.uint32), ov.view(np.uint32), err_msg=name)
        np.testing.assert_array_equal(i, oi, err_msg=name)
    with pytest.raises(IOError):
        app.obj_import(str(tmp_path / "missing.obj"))


def test_host_halton_and_frame_constants_match_oracle(built):
    """RayTracer::UpdateFrame of the host layer vs the oracle's restatement, byte for byte, over a FrameIndex wrap."""
    from raytracedggx_amd import app
    xy = app.halton(600)
    o = O.Oracle(16, 16, threads=1)
    for k in range(600):
        np.testing.assert_array_equal(xy[k], o.halton())
    for (W, H, ps, dt) in ((1920, 1080, (0, 0, 0, 1), 1 / 60), (640, 360, (0.0, 2.8, 0.0, 0.03), 0.0), (3840, 2160, (1, 2, 3, 0.5), 0.125)):
        frames = 260
        fc = app.frame_constants(W, H, frames, dt=dt, pos_scale=ps)
        o = O.Oracle(W, H, threads=1)
        o.set_pos_scale(ps)
        vp = O.camera_view_proj(W, H)
        for f in range(frames):
            o.update_frame((10, 10, -24), vp, dt)
            want = o.get_frame_constants()
            np.testing.assert_array_equal(fc[f][:704], want[:704], err_msg="frame %d" % f)
        assert fc[-1].view(np.uint32)[111] == (frames - 1) % 256      # CBGlobal::FrameIndex at byte 444
    # first frame: previous WVP defined as the current one; second frame: the first frame's
    fc = app.frame_constants(320, 180, 2).view(np.float32)
    np.testing.assert_array_equal(fc[0][32:64], fc[0][0:32])
    np.testing.assert_array_equal(fc[1][32:64], fc[0][0:32])


def test_headless_executable_reports_missing_gpu(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    exe = os.path.join(ROOT, "raytracedggx_amd", "RayTracedGGX")
    r = subprocess.run([exe, "-mesh", assets.path("triangle.obj"), "-env", assets.path("rnl_cross.dds"), "-width", "64", "-height", "64"],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "no HIP device" in r.stderr


def test_headless_executable_gpus_flag_starts_one_process_per_gpu(built):
    """-gpus N: the executable restarts itself once per GPU (host/Strips.cpp LaunchRanks) and returns the worst exit code.
    Without a GPU every rank must fail loudly -- there is no CPU path -- and so must the launcher."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    exe = os.path.join(ROOT, "raytracedggx_amd", "RayTracedGGX")
    r = subprocess.run([exe, "-gpus", "3", "-mesh", assets.path("triangle.obj"), "-env", assets.path("rnl_cross.dds"), "-width", "64", "-height", "64"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and r.stderr.count("no HIP device") >= 3, r.stderr
    r = subprocess.run([exe, "-gpus", "0"], capture_output=True, text=True)
    assert r.returncode == 1 and "-gpus: 1 .. 64" in r.stderr


def test_launcher_stops_a_hung_rank_and_refuses_under_a_profiler(built):
    """LaunchRanks (host/Strips.cpp): a rank that fails takes the others down after a short grace period -- a peer that waits for it inside
    RCCL would wait for ever --, and the launcher does not fork + exec at all when a profiler or preloaded GPU library is around
    (such a process has initialised the GPU before main; replacing it is not allowed on this pool)."""
    import time
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    exe = os.path.join(ROOT, "raytracedggx_amd", "RayTracedGGX")
    args = [exe, "-gpus", "2", "-mesh", assets.path("triangle.obj"), "-env", assets.path("rnl_cross.dds"), "-width", "64", "-height", "64"]
    t0 = time.time()
    r = subprocess.run(args, capture_output=True, text=True, timeout=60, env=dict(os.environ, RTGGX_DEBUG_RANK_SLEEP="1:40"))      # rank 0 fails at once (no GPU), rank 1 "hangs"
    assert r.returncode != 0 and "a rank failed" in r.stderr and time.time() - t0 < 20, r.stderr
    t0 = time.time()
    r = subprocess.run(args, capture_output=True, text=True, timeout=60, env=dict(os.environ, RTGGX_DEBUG_RANK_SLEEP="0:40", RTGGX_RANK_TIMEOUT="3"))     # (rank 1 fails; same path)
    assert r.returncode != 0 and time.time() - t0 < 20, r.stderr
    for var, val in (("HSA_TOOLS_LIB", "librocprofiler-sdk-tool.so"), ("LD_PRELOAD", "librocprofiler-sdk-does-not-exist.so")):      # (nothing is loaded: the runtime is never initialised, a missing preload is ignored by ld.so)
        r = subprocess.run(args, capture_output=True, text=True, timeout=60, env=dict(os.environ, **{var: val}))
        assert r.returncode == 2 and "refusing to start ranks" in r.stderr and "no HIP device" not in r.stderr, (var, r.stderr)


def test_cpp_strip_plans_equal_the_python_ones(built):
    """host/Strips.cpp (the executable's multi-GPU host) and raytracedggx_amd/strips.py (what bench.py drives) must cut the frame
    and pair the transfers identically: exchange plans for every rank of 2..8 strips over equal and uneven boundaries, balanced
    boundaries from random and from degenerate row costs, and the errors for frames too small to hold the strips."""
    import ctypes as C
    from raytracedggx_amd import app, strips
    L = app.load()
    L.rtggx_host_exchange_plan.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int]
    L.rtggx_host_balanced_bounds.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_uint32, C.c_double, C.c_void_p]
    rng = np.random.default_rng(5)
    def cpp_plan(H, rank, world, apron, bounds):
        ops = np.zeros((64, 5), np.int32)
        b = None if bounds is None else np.asarray(bounds, np.uint32)
        n = L.rtggx_host_exchange_plan(H, rank, world, apron, None if b is None else b.ctypes.data, ops.ctypes.data, 64)
        assert n >= 0
        return [("send" if o[0] else "recv", ("backbuffer", "history", "token")[o[1]], int(o[2]), int(o[3]), int(o[4])) for o in ops[:n]]
    for H in (272, 1080, 2160):
        for world in range(2, 9):
            cost = rng.random(H) * 100 + 1
            got = np.zeros(world + 1, np.uint32)
            for extra in (0.0, strips.gather_cost(1920, H, world, 0.02), cost.sum()):      # no gather cost, the default's, an absurd one (rank 0 is left its minimum)
                want = strips.balanced_bounds(cost, world, first_extra=extra)
                assert L.rtggx_host_balanced_bounds(cost.ctypes.data, H, world, strips.HISTORY_APRON, extra, got.ctypes.data) == 0
                assert got.tolist() == want
                if extra:
                    assert want[1] <= strips.balanced_bounds(cost, world)[1], "what rank 0 carries besides its rows makes its strip thinner"
            want = strips.balanced_bounds(cost, world)
            for bounds in (None, want):
                for apron in (strips.HISTORY_APRON, 30):
                    if any(b1 - b0 < apron for b0, b1 in zip(want[:-1], want[1:])) and bounds is not None:
                        continue
                    for rank in range(world):
                        assert cpp_plan(H, rank, world, apron, bounds) == strips.exchange_plan(H, rank, world, apron=apron, bounds=bounds)
    # all the cost in one row: the minimum strip height decides
    cost = np.zeros(360); cost[200] = 1.0
    got = np.zeros(7, np.uint32)
    assert L.rtggx_host_balanced_bounds(cost.ctypes.data, 360, 6, 18, 0.0, got.ctypes.data) == 0 and got.tolist() == strips.balanced_bounds(cost, 6)
    assert L.rtggx_host_balanced_bounds(cost.ctypes.data, 100, 6, 18, 0.0, got.ctypes.data) == -1 and b"cannot hold" in L.rtggx_app_last_error()
    with pytest.raises(ValueError):
        strips.balanced_bounds(cost[:100], 6)


def test_png_writer_and_image_diff_tool(built, tmp_path):
    """the host's PNG writer (the sample's screenshot container) read back by tools/imgdiff.py's own
    reader, byte for byte; the reader against PNGs with real scanline filters; the tool's verdicts and exit codes."""
    import ctypes
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import imgdiff
    L = ctypes.CDLL(os.path.join(ROOT, "raytracedggx_amd", "libRayTracedGGX.so"))
    rng = np.random.default_rng(3)
    for w, h, c in ((7, 5, 3), (300, 250, 4), (1, 1, 3), (256, 257, 3)):          # 300x250x4 needs several stored deflate blocks
        a = rng.integers(0, 256, (h, w, c), dtype=np.uint8)
        path = str(tmp_path / ("w%d.png" % w))
        assert L.rtggx_host_write_png(path.encode(), w, h, c, a.ctypes.data_as(ctypes.c_void_p)) == 0
        np.testing.assert_array_equal(imgdiff.load(path), a)
    assert L.rtggx_host_write_png(str(tmp_path / "bad.png").encode(), 4, 4, 2, None) != 0
    # filters 1-4: zlib-compressed PNGs from another encoder, when one is importable
    yy, xx = np.mgrid[0:97, 0:131]
    smooth = np.stack([(xx * 2) % 256, (yy * 3 + xx) % 256, ((xx * yy) // 7) % 256], axis=-1).astype(np.uint8)
    try:
        from PIL import Image
        Image.fromarray(smooth).save(str(tmp_path / "pil.png"))
        np.testing.assert_array_equal(imgdiff.load(str(tmp_path / "pil.png")), smooth)
    except ImportError:
        pass
    imgdiff.save_png(str(tmp_path / "a.png"), smooth)
    np.testing.assert_array_equal(imgdiff.load(str(tmp_path / "a.png")), smooth)
    b = smooth.copy(); b[10, 20, 1] ^= 1
    open(str(tmp_path / "b.ppm"), "wb").write(b"P6\n# comment\n131 97\n255\n" + b.tobytes())
    run = lambda *a: subprocess.run([sys.executable, os.path.join(ROOT, "tools", "imgdiff.py")] + list(a), capture_output=True, text=True)
    r = run(str(tmp_path / "a.png"), str(tmp_path / "b.ppm"), "--out", str(tmp_path / "d.png"))
    assert r.returncode == 0 and "max difference 1 codes" in r.stdout
    assert imgdiff.load(str(tmp_path / "d.png"))[10, 20, 1] == 255
    b[0, 0, 0] = (int(b[0, 0, 0]) + 9) % 256
    imgdiff.save_png(str(tmp_path / "c.png"), b)
    assert run(str(tmp_path / "a.png"), str(tmp_path / "c.png")).returncode == 1
    imgdiff.save_png(str(tmp_path / "e.png"), b[:50])
    assert run(str(tmp_path / "a.png"), str(tmp_path / "e.png")).returncode == 2


def test_camera_handlers_orbit_and_dolly(built):
    """OnLButtonDown / OnMouseMove / OnMouseWheel (RayTracedGGX.cpp:400-455), restated here with numpy: orbiting keeps the
    distance to the focus point and the focus on the view axis; the wheel scales the distance by 1 - dz/16; nothing
    moves while the button is up."""
    from raytracedggx_amd import app
    W, H = 1280, 720
    focus = np.array([0.0, 3.0, 0.0])
    eye0, view0 = app.camera(W, H, [])
    np.testing.assert_allclose(eye0, [10.0, 10.0, -24.0])
    d0 = np.linalg.norm(eye0 - focus)
    eye, view = app.camera(W, H, [(3, 500, 300)])                       # move without the button: ignored
    np.testing.assert_array_equal(eye, eye0); np.testing.assert_array_equal(view, view0)
    ev = [(1, 640, 360), (3, 600, 330), (3, 520, 345), (2, 0, 0), (3, 0, 0)]
    eye, view = app.camera(W, H, ev)
    assert abs(np.linalg.norm(eye - focus) - d0) < 2e-4 * d0 and np.linalg.norm(eye - eye0) > 1.0
    f = np.append(focus, 1.0) @ view.astype(np.float64)                  # the focus point in view space: on the +z axis
    assert abs(f[0]) < 1e-3 and abs(f[1]) < 1e-3 and abs(f[2] - d0) < 2e-4 * d0
    np.testing.assert_allclose(np.append(eye, 1.0) @ view.astype(np.float64), [0, 0, 0, 1], atol=2e-4)
    # the same orbit restated: view' = view * T(0,0,-len) * R(pitch, yaw) * T(0,0,len), one step per move
    def T(z):
        m = np.eye(4); m[3, 2] = z; return m
    def R(p, y):
        cp, sp, cy, sy = np.cos(p), np.sin(p), np.cos(y), np.sin(y)
        return np.array([[cy, 0, -sy, 0], [sp * sy, cp, sp * cy, 0], [cp * sy, -sp, cp * cy, 0], [0, 0, 0, 1]])
    v, e, last = view0.astype(np.float64), eye0.astype(np.float64), (640.0, 360.0)
    for x, y in ((600, 330), (520, 345)):
        dx, dy = last[0] - x, last[1] - y
        ln = np.linalg.norm(focus - e)
        v = v @ T(-ln) @ R(2 * np.pi * dy / H, 2 * np.pi * dx / W) @ T(ln)
        e = np.linalg.inv(v)[3, :3]; last = (x, y)
    np.testing.assert_allclose(view, v, atol=5e-5); np.testing.assert_allclose(eye, e, atol=5e-4)
    eye2, _ = app.camera(W, H, ev + [(4, 4.0, 0)])                       # wheel +4: a quarter closer
    assert abs(np.linalg.norm(eye2 - focus) - 0.75 * d0) < 1e-3


def test_bvh4_top_table_checker_on_a_synthetic_tree():
    """tests/bvh_checks.py: bvh4_top_check is what the GPU tests trust for the LDS table of the trace kernel (RTGGX_BUF_BVH4_TOP*).
    A small 4-wide tree by hand, its table for several capacities, and three ways of getting the table wrong."""
    import bvh_checks
    E = 0x7FFFFFFF
    n4 = np.zeros((8, 32), np.uint32)
    def node(i, refs):
        n4[i, :24] = np.arange(24) + 100 * i + 1
        n4[i, 24:28] = np.array(refs, np.int32).view(np.uint32)
    node(0, [2, 4, -1, E]); node(2, [6, -2, -3, E]); node(4, [-4, -5, E, E]); node(6, [-6, -7, -8, -9])
    order = [0, 2, 4, 6]
    def table(cap):
        used = order[:cap]
        t = n4[used].copy()
        for k in range(len(used)):
            r = t[k, 24:28].view(np.int32)
            for e in range(4):
                if 0 <= r[e] != E and r[e] in used:
                    r[e] = bvh_checks.TOP_FLAG | used.index(int(r[e]))
        return t
    for cap, want in ((1, 1), (2, 2), (3, 3), (4, 4), (16, 4)):
        assert bvh_checks.bvh4_top_check(n4, table(cap), 0, cap) == want
    bad = table(4); bad[1, 3] ^= 1                                   # a box bit
    with pytest.raises(AssertionError):
        bvh_checks.bvh4_top_check(n4, bad, 0, 4)
    bad = table(4); bad[0, 24] = 2                                   # a reference into the table left raw
    with pytest.raises(AssertionError):
        bvh_checks.bvh4_top_check(n4, bad, 0, 4)
    with pytest.raises(AssertionError):                              # too few entries for the capacity
        bvh_checks.bvh4_top_check(n4, table(2), 0, 4)
