"""Test/bench inputs: writes .obj / .dds files equivalent to the reference's assets from
tests/golden/assets.npz into a cache directory (the GPU box has no /root/reference)."""
import os
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_NPZ = os.path.join(_HERE, "golden", "assets.npz")
_CACHE = os.environ.get("RTGGX_ASSET_CACHE", os.path.join(tempfile.gettempdir(), "rtggx_assets_%d" % os.getuid()))

_SINGLE_TRIANGLE_OBJ = "v -1 0 0\nv 1 0 0\nv 0 2 0\nf 1 2 3\n"


def _write_atomic(path, data, mode):
    tmp = "%s.%d.tmp" % (path, os.getpid())
    with open(tmp, mode) as f:
        f.write(data)
    os.replace(tmp, path)


def asset_dir():
    """Directory laid out like the reference's Bin/: Assets/bunny.obj, Assets/dragon.obj, Assets/TuringBowl.obj, Assets/rnl_cross.dds."""
    d = os.path.join(_CACHE, "Assets")
    os.makedirs(d, exist_ok=True)
    need = [n for n in ("bunny.obj", "dragon.obj", "TuringBowl.obj", "rnl_cross.dds", "triangle.obj") if not os.path.exists(os.path.join(d, n))]
    if need:
        z = np.load(_NPZ)
        for name in ("bunny", "dragon", "TuringBowl"):
            if name + ".obj" in need:
                v, f = z[name + "_v"], z[name + "_f"]
                lines = ["v %.9g %.9g %.9g" % tuple(float(x) for x in row) for row in v]
                if name + "_vn" in z.files:          # "f a//na b//nb c//nc" (TuringBowl: per-corner normals, vertex splitting)
                    lines += ["vn %.9g %.9g %.9g" % tuple(float(x) for x in row) for row in z[name + "_vn"]]
                    lines += ["f %d//%d %d//%d %d//%d" % tuple(int(x) for x in row.reshape(-1)) for row in f]
                else:
                    lines += ["f %d %d %d" % tuple(int(x) for x in row) for row in f]
                _write_atomic(os.path.join(d, name + ".obj"), "\n".join(lines) + "\n", "w")
        if "rnl_cross.dds" in need:
            _write_atomic(os.path.join(d, "rnl_cross.dds"), z["rnl_cross_dds"].tobytes(), "wb")
        if "triangle.obj" in need:
            _write_atomic(os.path.join(d, "triangle.obj"), _SINGLE_TRIANGLE_OBJ, "w")
    return os.path.dirname(d)


def path(name):
    return os.path.join(asset_dir(), "Assets", name)


def constant_env_rgba16f(value=1.0):
    """6 faces x 1 texel of (v,v,v,1) as RGBA16F: the constant environment of config C1."""
    t = np.zeros((6, 4), np.float16)
    t[:, :3] = value
    t[:, 3] = 1.0
    return t.view(np.uint16)


def fnv1a32(data):
    h = 0x811C9DC5
    for b in bytes(data):
        h = ((h ^ b) * 0x01000193) & 0xFFFFFFFF
    return h
