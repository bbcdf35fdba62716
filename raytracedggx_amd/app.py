"""Python driver over the C++ host layer (libRayTracedGGX.so): the frame entry is the reference's
RayTracedGGX::OnUpdate / OnRender, the constructor takes the reference's command line
(`-mesh <obj> x y z s`, `-env <dds>`; RayTracedGGX.cpp:462-511) plus the headless extensions.
"""
import ctypes as C
import os

import numpy as np

from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_HERE, "libRayTracedGGX.so")
HOST_EXPORTS = ["rtggx_app_last_error", "rtggx_app_create", "rtggx_app_destroy", "rtggx_app_on_update", "rtggx_app_on_render",
                "rtggx_app_on_key_up", "rtggx_app_set_time_step", "rtggx_app_context", "rtggx_app_size",
                "rtggx_app_frame_constants", "rtggx_app_save_image", "rtggx_host_obj_import", "rtggx_host_obj_copy",
                "rtggx_host_halton", "rtggx_host_frame_constants", "rtggx_host_write_png", "rtggx_host_camera",
                "rtggx_app_on_lbutton_down", "rtggx_app_on_lbutton_up", "rtggx_app_on_mouse_move", "rtggx_app_on_mouse_wheel", "rtggx_app_load_track",
                "rtggx_host_exchange_plan", "rtggx_host_balanced_bounds", "rtggx_app_set_dump_prefix", "rtggx_app_last_screen_shot"]

_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise ImportError("libRayTracedGGX.so is not built: run __graft_entry__.build() or `make -C raytracedggx_amd`")
        capi.load()  # librtggx.so first (same directory, also found through rpath)
        L = C.CDLL(HOST_LIB_PATH)
        L.rtggx_app_last_error.restype = C.c_char_p
        L.rtggx_app_create.restype = C.c_void_p
        L.rtggx_app_create.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
        L.rtggx_app_context.restype = C.c_void_p
        for n in ("rtggx_app_destroy", "rtggx_app_on_update", "rtggx_app_on_render"):
            getattr(L, n).argtypes = [C.c_void_p]
            getattr(L, n).restype = None
        L.rtggx_app_context.argtypes = [C.c_void_p]
        L.rtggx_app_on_key_up.argtypes = [C.c_void_p, C.c_int]
        L.rtggx_app_set_time_step.argtypes = [C.c_void_p, C.c_float]
        L.rtggx_app_size.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.rtggx_app_frame_constants.argtypes = [C.c_void_p, C.c_void_p]
        L.rtggx_app_save_image.argtypes = [C.c_void_p, C.c_char_p]
        L.rtggx_app_set_dump_prefix.argtypes = [C.c_void_p, C.c_char_p]
        L.rtggx_app_last_screen_shot.argtypes = [C.c_void_p]
        L.rtggx_app_last_screen_shot.restype = C.c_char_p
        L.rtggx_host_obj_import.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_void_p]
        L.rtggx_host_obj_copy.argtypes = [C.c_void_p, C.c_void_p]
        L.rtggx_host_halton.argtypes = [C.c_uint32, C.c_void_p]
        L.rtggx_host_frame_constants.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_uint32, C.c_void_p]
        for n in ("rtggx_app_on_lbutton_down", "rtggx_app_on_lbutton_up", "rtggx_app_on_mouse_move"):
            getattr(L, n).argtypes = [C.c_void_p, C.c_float, C.c_float]
            getattr(L, n).restype = None
        L.rtggx_app_on_mouse_wheel.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
        L.rtggx_app_on_mouse_wheel.restype = None
        L.rtggx_app_load_track.argtypes = [C.c_void_p, C.c_char_p]
        L.rtggx_host_camera.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.rtggx_host_camera.restype = None
        L.rtggx_host_write_png.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        _lib = L
    return _lib


def obj_import(path):
    """ObjLoader::Import of the product's host layer -> (verts[nv,6], indices[ni], aabb[6])."""
    L = load()
    nv, ni = C.c_uint32(), C.c_uint32()
    aabb = np.zeros(6, np.float32)
    if L.rtggx_host_obj_import(path.encode(), C.byref(nv), C.byref(ni), aabb.ctypes.data_as(C.c_void_p)) != 0:
        raise IOError(L.rtggx_app_last_error().decode())
    v = np.zeros((nv.value, 6), np.float32)
    i = np.zeros(ni.value, np.uint32)
    L.rtggx_host_obj_copy(v.ctypes.data_as(C.c_void_p), i.ctypes.data_as(C.c_void_p))
    return v, i, aabb


def camera(width, height, events):
    """The camera handlers on their own: events = [(type, a, b)], type 1 down, 2 up, 3 move, 4 wheel -> (eye[3], view[4, 4])."""
    ev = np.ascontiguousarray(np.asarray(events, np.float32).reshape(-1, 3))
    eye, view = np.zeros(3, np.float32), np.zeros((4, 4), np.float32)
    load().rtggx_host_camera(C.c_uint32(width), C.c_uint32(height), ev.ctypes.data_as(C.c_void_p), C.c_uint32(len(ev)),
                             eye.ctypes.data_as(C.c_void_p), view.ctypes.data_as(C.c_void_p))
    return eye, view


def halton(n):
    xy = np.zeros((n, 2), np.float32)
    load().rtggx_host_halton(n, xy.ctypes.data_as(C.c_void_p))
    return xy


def frame_constants(width, height, frames, dt=1.0 / 60.0, pos_scale=(0, 0, 0, 1), eye=(10.0, 10.0, -24.0), focus=(0.0, 3.0, 0.0)):
    """RayTracer::UpdateFrame of the host layer for `frames` consecutive frames -> uint8[frames, 768]."""
    out = np.zeros((frames, 768), np.uint8)
    ps, e, f = (np.asarray(a, np.float32) for a in (pos_scale, eye, focus))
    load().rtggx_host_frame_constants(width, height, ps.ctypes.data_as(C.c_void_p), e.ctypes.data_as(C.c_void_p),
                                      f.ctypes.data_as(C.c_void_p), dt, frames, out.ctypes.data_as(C.c_void_p))
    return out


class BorrowedContext(capi.Context):
    """capi.Context view of the rtggx_context owned by a RayTracedGGX application object."""

    def __init__(self, handle, width, height):  # pylint: disable=super-init-not-called
        self.L = capi.load()
        self.W, self.H = width, height
        self.h = C.c_void_p(handle)

    def close(self):
        self.h = None


class RayTracedGGX:
    """The reference's application object, headless: RayTracedGGX(args).OnUpdate()/OnRender()."""

    def __init__(self, args):
        self.L = load()
        argv = [b"RayTracedGGX"] + [str(a).encode() for a in args]
        arr = (C.c_char_p * len(argv))(*argv)
        self.h = self.L.rtggx_app_create(len(argv), arr)
        if not self.h:
            raise capi.RtggxError("RayTracedGGX::OnInit failed: " + self.L.rtggx_app_last_error().decode())
        w, h = C.c_uint32(), C.c_uint32()
        self.L.rtggx_app_size(self.h, C.byref(w), C.byref(h))
        self.width, self.height = w.value, h.value
        self.context = BorrowedContext(self.L.rtggx_app_context(self.h), self.width, self.height)

    def OnUpdate(self):
        self.L.rtggx_app_on_update(self.h)

    def OnRender(self):
        self.L.rtggx_app_on_render(self.h)

    def OnKeyUp(self, key):
        self.L.rtggx_app_on_key_up(self.h, int(key))

    # the sample's camera interactions (RayTracedGGX.cpp:400-455), positions in pixels
    def OnLButtonDown(self, x, y):
        self.L.rtggx_app_on_lbutton_down(self.h, x, y)

    def OnLButtonUp(self, x, y):
        self.L.rtggx_app_on_lbutton_up(self.h, x, y)

    def OnMouseMove(self, x, y):
        self.L.rtggx_app_on_mouse_move(self.h, x, y)

    def OnMouseWheel(self, dz, x=0.0, y=0.0):
        self.L.rtggx_app_on_mouse_wheel(self.h, dz, x, y)

    def load_track(self, path):
        return self.L.rtggx_app_load_track(self.h, path.encode()) == 0

    def set_time_step(self, dt):
        self.L.rtggx_app_set_time_step(self.h, dt)

    def frame_constants(self):
        b = np.zeros(768, np.uint8)
        self.L.rtggx_app_frame_constants(self.h, b.ctypes.data_as(C.c_void_p))
        return b

    def save_image(self, path):
        return self.L.rtggx_app_save_image(self.h, path.encode()) == 0

    def set_dump_prefix(self, prefix):
        """Where [F11] screen shots go: <prefix>_f<frame>.png (the -dump flag)."""
        self.L.rtggx_app_set_dump_prefix(self.h, prefix.encode())

    def last_screen_shot(self):
        """The file the most recent [F11] (key code 0x7A, `key F11` in a -track script) wrote; '' if none."""
        return self.L.rtggx_app_last_screen_shot(self.h).decode()

    def OnDestroy(self):
        if self.h:
            self.context.close()
            self.L.rtggx_app_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.OnDestroy()
        except Exception:
            pass
