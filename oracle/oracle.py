"""ORACLE -- TEST INFRASTRUCTURE ONLY.

ctypes wrapper of oracle/_build/liboracle.so (the scalar C++ restatement of the reference's hot
path).  Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only; the
product package raytracedggx_amd never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")

BUF_VISIBILITY, BUF_DEPTH, BUF_NORMAL, BUF_ROUGH_METAL, BUF_VELOCITY, BUF_RT_REFL, BUF_RT_DIFF, \
    BUF_TSS0, BUF_TSS1, BUF_FLT_RFL, BUF_FLT_DFF, BUF_BACKBUFFER, BUF_SH_COEFFS = range(13)
_DTYPES = {BUF_VISIBILITY: np.uint32, BUF_DEPTH: np.uint32, BUF_NORMAL: np.uint32, BUF_ROUGH_METAL: np.uint16,
           BUF_VELOCITY: np.uint32, BUF_RT_REFL: np.uint32, BUF_RT_DIFF: np.uint32, BUF_TSS0: np.uint64,
           BUF_TSS1: np.uint64, BUF_FLT_RFL: np.uint64, BUF_FLT_DFF: np.uint64, BUF_BACKBUFFER: np.uint32,
           BUF_SH_COEFFS: np.float32}


def build(force=False):
    """Compile the oracle (make -C oracle).  Building the checker is not using it."""
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
            for f in os.listdir(_HERE) if f.endswith((".h", ".cpp"))):
        subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_buffer.restype = C.c_void_p
        L.orc_buffer.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
        L.orc_ray_trace.restype = C.c_uint64
        L.orc_env_texel_count.restype = C.c_uint64
        L.orc_rng.restype = C.c_uint32
        L.orc_rng.argtypes = [C.c_uint32]
        L.orc_pack_r11g11b10f.restype = C.c_uint32
        L.orc_pack_r10g10b10a2.restype = C.c_uint32
        L.orc_f32_to_f16.restype = C.c_uint16
        L.orc_f32_to_f16.argtypes = [C.c_float]
        L.orc_f16_to_f32.restype = C.c_float
        L.orc_f16_to_f32.argtypes = [C.c_uint16]
        L.orc_get_parity.restype = C.c_uint32
        for name in ("orc_destroy", "orc_set_threads", "orc_set_mesh", "orc_set_pos_scale", "orc_set_metallic", "orc_set_sampler",
                     "orc_set_material", "orc_set_env_dds", "orc_set_env_rgba16f", "orc_env_texel_count", "orc_env_info",
                     "orc_env_copy", "orc_build_as", "orc_set_bvh", "orc_bvh_info", "orc_bvh_copy", "orc_update_frame",
                     "orc_set_frame_constants", "orc_get_frame_constants", "orc_halton", "orc_update_as",
                     "orc_get_inv_worlds", "orc_transform_sh", "orc_set_sh", "orc_render_visibility", "orc_set_visibility",
                     "orc_ray_trace", "orc_denoise", "orc_tone_map", "orc_flip_parity", "orc_get_parity", "orc_trace_rays",
                     "orc_environment"):
            fn = getattr(L, name)
            if fn.argtypes is None:
                fn.argtypes = None  # first arg is the handle: always pass C.c_void_p explicitly
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.c_void_p)


def obj_import(path):
    """ObjLoader::Import restatement -> (verts[nv,6] float32, indices[ni] uint32, aabb[6])."""
    L = lib()
    nv, ni = C.c_uint32(), C.c_uint32()
    aabb = np.zeros(6, np.float32)
    if L.orc_obj_import(path.encode(), C.byref(nv), C.byref(ni), _fp(aabb)) != 0:
        raise IOError("oracle: cannot import " + path)
    verts = np.zeros((nv.value, 6), np.float32)
    idx = np.zeros(ni.value, np.uint32)
    L.orc_obj_copy(_fp(verts), _fp(idx))
    return verts, idx, aabb


def bc6h_decode_block(block16, signed=False):
    out = np.zeros((16, 3), np.uint16)
    b = np.frombuffer(bytes(block16), np.uint8).copy()
    (lib().orc_bc6h_decode_block_sf16 if signed else lib().orc_bc6h_decode_block)(_fp(b), _fp(out))
    return out


def camera_view_proj(width, height, eye=(10.0, 10.0, -24.0), focus=(0.0, 3.0, 0.0)):
    vp = np.zeros((4, 4), np.float32)
    lib().orc_camera_view_proj(C.c_uint32(width), C.c_uint32(height), _fp(np.asarray(eye, np.float32)),
                               _fp(np.asarray(focus, np.float32)), _fp(vp))
    return vp


class Oracle:
    """One scene + render-target set; methods follow RayTracer / Denoiser of the reference."""

    def __init__(self, width, height, threads=None):
        self.L = lib()
        self.W, self.H = width, height
        self.h = C.c_void_p(self.L.orc_create(width, height))
        if not self.h:
            raise RuntimeError("orc_create failed")
        self.set_threads(threads if threads else min(os.cpu_count() or 1, 16))

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_threads(self, n):
        self.threads = int(n)
        self.L.orc_set_threads(self.h, C.c_int(int(n)))

    # --- inputs
    def set_mesh(self, slot, verts, idx):
        verts = np.ascontiguousarray(verts, np.float32)
        idx = np.ascontiguousarray(idx, np.uint32)
        self.L.orc_set_mesh(self.h, C.c_uint32(slot), _fp(verts), C.c_uint32(verts.shape[0]), _fp(idx), C.c_uint32(idx.size))

    def set_pos_scale(self, ps):
        self.L.orc_set_pos_scale(self.h, _fp(np.asarray(ps, np.float32)))

    def set_sampler(self, vndf):
        self.L.orc_set_sampler(self.h, C.c_int(1 if vndf else 0))

    def set_normal_weight(self, variant):
        """How the spatial filters' pow(dot(N, Nc), 512 | 32) is evaluated, process-wide: "exact" (double, rounded once; the default),
        "libm" (fp32 dot product, std::pow in fp32) or "squared" (round 3's nine / five squarings)."""
        self.L.orc_set_normal_weight(C.c_int({"exact": 0, "libm": 1, "squared": 2}[variant]))

    def set_metallic(self, mesh, m):
        self.L.orc_set_metallic(self.h, C.c_uint32(mesh), C.c_float(m))

    def set_material(self, mesh, base_color, rough, metal):
        self.L.orc_set_material(self.h, C.c_uint32(mesh), _fp(np.asarray(base_color, np.float32)), C.c_float(rough), C.c_float(metal))

    def set_env_dds(self, path):
        if self.L.orc_set_env_dds(self.h, path.encode()) != 0:
            raise IOError("oracle: cannot load " + path)

    def set_env_rgba16f(self, size, mips, texels_u16):
        t = np.ascontiguousarray(texels_u16, np.uint16)
        self.L.orc_set_env_rgba16f(self.h, C.c_uint32(size), C.c_uint32(mips), _fp(t))

    def env_texels(self):
        n = self.L.orc_env_texel_count(self.h)
        out = np.zeros((n, 4), np.uint16)
        self.L.orc_env_copy(self.h, _fp(out))
        size, mips = C.c_uint32(), C.c_uint32()
        self.L.orc_env_info(self.h, C.byref(size), C.byref(mips))
        return size.value, mips.value, out

    # --- acceleration structure
    def build_as(self):
        self.L.orc_build_as(self.h)

    def set_bvh(self, slot, nodes, tris, root):
        nodes = np.ascontiguousarray(nodes)
        tris = np.ascontiguousarray(tris)
        self.L.orc_set_bvh(self.h, C.c_uint32(slot), _fp(nodes), C.c_uint32(nodes.nbytes // 64), _fp(tris),
                           C.c_uint32(tris.nbytes // 64), C.c_int32(root))

    def get_bvh(self, slot):
        nn, nt, root = C.c_uint32(), C.c_uint32(), C.c_int32()
        self.L.orc_bvh_info(self.h, C.c_uint32(slot), C.byref(nn), C.byref(nt), C.byref(root))
        nodes = np.zeros((nn.value, 16), np.uint32)
        tris = np.zeros((nt.value, 16), np.uint32)
        self.L.orc_bvh_copy(self.h, C.c_uint32(slot), _fp(nodes), _fp(tris))
        return nodes, tris, root.value

    # --- per frame
    def update_frame(self, eye, view_proj, dt):
        self.L.orc_update_frame(self.h, _fp(np.asarray(eye, np.float32)), _fp(np.ascontiguousarray(view_proj, np.float32)), C.c_float(dt))

    def set_frame_constants(self, fc_bytes):
        b = np.frombuffer(bytes(fc_bytes), np.uint8).copy()
        assert b.size == 768
        self.L.orc_set_frame_constants(self.h, _fp(b))

    def get_frame_constants(self):
        b = np.zeros(768, np.uint8)
        self.L.orc_get_frame_constants(self.h, _fp(b))
        return b

    def halton(self):
        xy = np.zeros(2, np.float32)
        self.L.orc_halton(self.h, _fp(xy))
        return xy

    def update_as(self):
        self.L.orc_update_as(self.h)

    def inv_worlds(self):
        m = np.zeros((2, 4, 4), np.float32)
        self.L.orc_get_inv_worlds(self.h, _fp(m))
        return m

    def transform_sh(self):
        self.L.orc_transform_sh(self.h)

    def set_sh(self, sh):
        self.L.orc_set_sh(self.h, _fp(np.ascontiguousarray(sh, np.float32)))

    def render_visibility(self):
        self.L.orc_render_visibility(self.h)

    def set_visibility(self, vis, depth):
        self.L.orc_set_visibility(self.h, _fp(np.ascontiguousarray(vis, np.uint32)), _fp(np.ascontiguousarray(depth, np.uint32)))

    def ray_trace(self):
        return int(self.L.orc_ray_trace(self.h))

    def denoise(self):
        self.L.orc_denoise(self.h)

    def tone_map(self):
        self.L.orc_tone_map(self.h)

    def render(self, eye, view_proj, dt):
        """OnUpdate + OnRender of the reference for one frame; returns the non-degenerate ray count."""
        self.update_frame(eye, view_proj, dt)
        self.update_as()
        self.render_visibility()
        rays = self.ray_trace()
        self.denoise()
        self.tone_map()
        return rays

    # --- outputs
    def buffer(self, bid, copy=True):
        n = C.c_uint64()
        p = self.L.orc_buffer(self.h, C.c_int(bid), C.byref(n))
        dt = np.dtype(_DTYPES[bid])
        arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n.value,)).view(dt)
        if bid != BUF_SH_COEFFS:
            arr = arr.reshape(self.H, self.W)
        else:
            arr = arr.reshape(9, 3)
        return arr.copy() if copy else arr

    def parity(self):
        return int(self.L.orc_get_parity(self.h))

    def trace_rays(self, rays, brute=False):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        out = np.zeros((rays.shape[0], 6), np.float32)
        self.L.orc_trace_rays(self.h, _fp(rays), C.c_uint32(rays.shape[0]), C.c_int(1 if brute else 0), _fp(out))
        return {"t": out[:, 0].copy(), "inst": out[:, 1].copy().view(np.uint32), "prim": out[:, 2].copy().view(np.uint32),
                "b1": out[:, 3].copy(), "b2": out[:, 4].copy(), "valid": out[:, 5] > 0.5}

    def environment(self, direction, level):
        rgb = np.zeros(3, np.float32)
        self.L.orc_environment(self.h, _fp(np.asarray(direction, np.float32)), C.c_float(level), _fp(rgb))
        return rgb


# ---- numpy-side format helpers (decode packed words for tolerance comparisons) ----------------------
def unpack_r11g11b10f(words):
    w = np.asarray(words, np.uint32)

    def uf(v, mb):
        e = (v >> mb).astype(np.int32)
        m = (v & ((1 << mb) - 1)).astype(np.float64)
        den = m * 2.0 ** (-14 - mb)
        nor = (1.0 + m / (1 << mb)) * np.exp2((e - 15).astype(np.float64))
        out = np.where(e == 0, den, nor)
        out = np.where(e == 31, np.where(m == 0, np.inf, np.nan), out)
        return out.astype(np.float32)
    return np.stack([uf(w & 0x7FF, 6), uf((w >> 11) & 0x7FF, 6), uf(w >> 22, 5)], axis=-1)


def unpack_rgba16f(words):
    w = np.ascontiguousarray(words, np.uint64)
    return w.view(np.float16).reshape(w.shape + (4,)).astype(np.float32)


def unpack_rgba8(words):
    w = np.ascontiguousarray(words, np.uint32)
    return w.view(np.uint8).reshape(w.shape + (4,))
