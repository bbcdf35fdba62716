"""ctypes binding of librtggx.so -- the C ABI declared in include/rtggx.h.

The library is the product: hand-written HIP for gfx950 behind `extern "C"` entry points.  There is
no CPU fallback; loading fails loudly when the shared object is missing, and rtggx_create fails
when no HIP device is present.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librtggx.so")

# buffer ids (rtggx.h)
BUF_VISIBILITY, BUF_DEPTH, BUF_NORMAL, BUF_ROUGH_METAL, BUF_VELOCITY, BUF_RT_REFL, BUF_RT_DIFF, BUF_TSS0, BUF_TSS1, \
    BUF_FLT_RFL, BUF_FLT_DFF, BUF_BACKBUFFER, BUF_SH_COEFFS, BUF_BVH_NODES0, BUF_BVH_TRIS0, BUF_BVH_NODES1, BUF_BVH_TRIS1, \
    BUF_TLAS, BUF_ENV, BUF_BVH4_NODES0, BUF_BVH4_NODES1, BUF_BIN_WORK, BUF_BVH4_TOP0, BUF_BVH4_TOP1, BUF_EXCHANGE_TOKENS = range(25)
MAX_PEERS, IPC_HANDLE_BYTES = 16, 64
FORMAT_RGBA32F, FORMAT_RGBA16F, FORMAT_BC6H_UF16, FORMAT_BC6H_SF16 = 2, 10, 95, 96

_BUF_DTYPE = {BUF_VISIBILITY: np.uint32, BUF_DEPTH: np.uint32, BUF_NORMAL: np.uint32, BUF_ROUGH_METAL: np.uint16,
              BUF_VELOCITY: np.uint32, BUF_RT_REFL: np.uint32, BUF_RT_DIFF: np.uint32, BUF_TSS0: np.uint64, BUF_TSS1: np.uint64,
              BUF_FLT_RFL: np.uint64, BUF_FLT_DFF: np.uint64, BUF_BACKBUFFER: np.uint32, BUF_SH_COEFFS: np.float32,
              BUF_BVH_NODES0: np.uint32, BUF_BVH_TRIS0: np.uint32, BUF_BVH_NODES1: np.uint32, BUF_BVH_TRIS1: np.uint32,
              BUF_TLAS: np.float32, BUF_ENV: np.uint16, BUF_BVH4_NODES0: np.uint32, BUF_BVH4_NODES1: np.uint32, BUF_BIN_WORK: np.uint32,
              BUF_BVH4_TOP0: np.uint32, BUF_BVH4_TOP1: np.uint32, BUF_EXCHANGE_TOKENS: np.uint32}

EXPORTS = ["rtggx_last_error", "rtggx_create", "rtggx_destroy", "rtggx_set_strip", "rtggx_set_stream", "rtggx_set_mesh",
           "rtggx_set_env", "rtggx_set_material", "rtggx_set_metallic", "rtggx_build_as", "rtggx_update_frame", "rtggx_update_as",
           "rtggx_transform_sh", "rtggx_render_visibility", "rtggx_ray_trace", "rtggx_denoise", "rtggx_tone_map", "rtggx_sync",
           "rtggx_ray_count", "rtggx_get_timings", "rtggx_enable_timing", "rtggx_buffer_size", "rtggx_readback", "rtggx_buffer_ptr",
           "rtggx_upload", "rtggx_frame_parity", "rtggx_bvh_root", "rtggx_trace_rays", "rtggx_ray_total", "rtggx_kernel_times", "rtggx_debug_counters", "rtggx_debug_trace_split", "rtggx_debug_trace_residency", "rtggx_get_stream", "rtggx_set_history_peers", "rtggx_history_ipc_export", "rtggx_history_ipc_open",
           "rtggx_set_async_compute", "rtggx_set_history_apron", "rtggx_history_overreach", "rtggx_copy_bandwidth", "rtggx_refit_as", "rtggx_refit_as_device", "rtggx_refit_stats", "rtggx_set_refit_policy", "rtggx_set_sampler", "rtggx_debug_fuse_tone_map", "rtggx_debug_placement", "rtggx_debug_tile_words", "rtggx_debug_collapse_weights", "rtggx_debug_fence_wait", "rtggx_debug_shader_clock"]


class Timings(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("update_as", "visibility", "ray_trace", "spatial_refl_h", "spatial_refl_v",
                                         "spatial_diff_h", "spatial_diff_v", "temporal", "tone_map", "frame", "ray_trace_kernel")]


_lib = None


def load():
    """Load librtggx.so; raises if the HIP library has not been built (no fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("librtggx.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C raytracedggx_amd` (hipcc, gfx950). There is no CPU path.")
    L = C.CDLL(LIB_PATH)
    L.rtggx_last_error.restype = C.c_char_p
    vp = C.c_void_p
    L.rtggx_create.argtypes = [C.POINTER(vp), C.c_uint32, C.c_uint32, C.c_int]
    L.rtggx_destroy.argtypes = [vp]
    L.rtggx_destroy.restype = None
    L.rtggx_set_strip.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.rtggx_set_stream.argtypes = [vp, vp]
    L.rtggx_set_async_compute.argtypes = [vp, C.c_int]
    L.rtggx_set_history_apron.argtypes = [vp, C.c_uint32]
    L.rtggx_debug_shader_clock.argtypes = [vp, C.POINTER(C.c_double)]
    L.rtggx_refit_as.argtypes = [vp, C.c_uint32, vp, C.c_uint32]
    L.rtggx_set_refit_policy.argtypes = [vp, C.c_float, C.c_uint32]
    L.rtggx_refit_stats.argtypes = [vp, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.rtggx_copy_bandwidth.argtypes = [vp, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
    L.rtggx_history_overreach.argtypes = [vp, C.POINTER(C.c_uint32), C.c_int]
    L.rtggx_set_mesh.argtypes = [vp, C.c_uint32, vp, C.c_uint32, vp, C.c_uint32]
    L.rtggx_set_env.argtypes = [vp, C.c_int, C.c_uint32, C.c_uint32, vp, C.c_size_t]
    L.rtggx_set_material.argtypes = [vp, C.c_uint32, vp, C.c_float, C.c_float]
    L.rtggx_set_metallic.argtypes = [vp, C.c_uint32, C.c_float]
    for n in ("rtggx_build_as", "rtggx_update_as", "rtggx_transform_sh", "rtggx_render_visibility", "rtggx_ray_trace",
              "rtggx_tone_map", "rtggx_sync"):
        getattr(L, n).argtypes = [vp]
    L.rtggx_update_frame.argtypes = [vp, vp]
    L.rtggx_denoise.argtypes = [vp, C.c_int]
    L.rtggx_ray_count.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.rtggx_get_timings.argtypes = [vp, C.POINTER(Timings)]
    L.rtggx_enable_timing.argtypes = [vp, C.c_int]
    L.rtggx_buffer_size.argtypes = [vp, C.c_int, C.POINTER(C.c_size_t)]
    L.rtggx_readback.argtypes = [vp, C.c_int, vp, C.c_size_t]
    L.rtggx_buffer_ptr.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.rtggx_upload.argtypes = [vp, C.c_int, vp, C.c_size_t]
    L.rtggx_frame_parity.argtypes = [vp, C.POINTER(C.c_uint32)]
    L.rtggx_bvh_root.argtypes = [vp, C.c_uint32, C.POINTER(C.c_int32)]
    L.rtggx_trace_rays.argtypes = [vp, vp, C.c_uint32, vp]
    L.rtggx_ray_total.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int]
    L.rtggx_kernel_times.argtypes = [vp, vp, C.c_uint32, C.POINTER(C.c_uint32)]
    _lib = L
    return L


class RtggxError(RuntimeError):
    pass


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """One rtggx_context: scene + render targets on one GPU.  Method names follow rtggx.h."""

    def __init__(self, width, height, device=0):
        self.L = load()
        self.W, self.H = int(width), int(height)
        h = C.c_void_p()
        self._check(self.L.rtggx_create(C.byref(h), self.W, self.H, int(device)))
        self.h = h

    def _check(self, rc):
        if rc != 0:
            raise RtggxError("librtggx: %s (code %d)" % (self.L.rtggx_last_error().decode(), rc))

    def close(self):
        if getattr(self, "h", None):
            self.L.rtggx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_strip(self, row_begin, row_end):
        self._check(self.L.rtggx_set_strip(self.h, row_begin, row_end))

    def set_stream(self, stream_handle):
        self._check(self.L.rtggx_set_stream(self.h, C.c_void_p(stream_handle)))

    def shader_clock_mhz(self):
        m = C.c_double()
        self._check(self.L.rtggx_debug_shader_clock(self.h, C.byref(m)))
        return m.value

    def copy_bandwidth(self, nbytes=1 << 30, iterations=8):
        """GB/s (read + written) of a float4 copy kernel over two buffers of nbytes each: the attainable HBM peak on this box."""
        g = C.c_double()
        self._check(self.L.rtggx_copy_bandwidth(self.h, int(nbytes), int(iterations), C.byref(g)))
        return g.value

    def set_history_apron(self, rows):
        self._check(self.L.rtggx_set_history_apron(self.h, int(rows)))

    def history_overreach(self, reset=True):
        """Rows by which a history tap read beyond the delivered apron since the last reset (strips; 0 = exact)."""
        n = C.c_uint32()
        self._check(self.L.rtggx_history_overreach(self.h, C.byref(n), int(reset)))
        return n.value

    def set_async_compute(self, enable):
        self._check(self.L.rtggx_set_async_compute(self.h, int(bool(enable))))

    def set_mesh(self, slot, verts, indices):
        v = np.ascontiguousarray(verts, np.float32).reshape(-1, 6)
        i = np.ascontiguousarray(indices, np.uint32).reshape(-1)
        self._check(self.L.rtggx_set_mesh(self.h, slot, _p(v), v.shape[0], _p(i), i.size))

    def refit_as(self, slot, verts):
        """New vertex positions / normals for an unchanged topology: staged now, uploaded and refitted on stream B by the next frame."""
        v = np.ascontiguousarray(verts, np.float32).reshape(-1, 6)
        self._check(self.L.rtggx_refit_as(self.h, slot, _p(v), v.shape[0]))

    def refit_as_device(self, slot, device_ptr, num_verts, stream=0):
        """New vertices from DEVICE memory (a mesh animated on the GPU): ordered on `stream` like a hipMemcpyAsync out of the buffer."""
        self.L.rtggx_refit_as_device.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
        self._check(self.L.rtggx_refit_as_device(self.h, slot, C.c_void_p(int(device_ptr)), int(num_verts), C.c_void_p(int(stream) or None)))

    def fence_wait(self, reset=True):
        """(us the host has waited at the frames-in-flight fence, frames that waited) since the last reset (diagnostic)."""
        us, n = C.c_double(0.0), C.c_uint32(0)
        self.L.rtggx_debug_fence_wait.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint32), C.c_int]
        self._check(self.L.rtggx_debug_fence_wait(self.h, C.byref(us), C.byref(n), 1 if reset else 0))
        return us.value, n.value

    def fuse_tone_map(self, mode):
        """True / False: the temporal pass always / never tone-maps its result as well; None: the library's choice (small launches) (diagnostic)."""
        self._check(self.L.rtggx_debug_fuse_tone_map(self.h, -1 if mode is None else 1 if mode else 0))

    def tile_words(self, enable):
        """False: every 16x16 tile of the visibility target counts as drawn (rounds 1-3); True: the tiles' words decide (diagnostic)."""
        self._check(self.L.rtggx_debug_tile_words(self.h, 1 if enable else 0))

    def placement(self, force_small=-1):
        """Pins the `small launch` fact of the stream placement (0 / 1; -1: by the ray count) and returns the key and placement of the most
        recent ray_trace: ({small, strip, deforming, diffuse, caller_stream}, {gen, trace, shade: "main" | "B" | "C" | "R", frames_in_flight})."""
        k, w = C.c_uint32(), C.c_uint32()
        self.L.rtggx_debug_placement.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        self._check(self.L.rtggx_debug_placement(self.h, int(force_small), C.byref(k), C.byref(w)))
        names = ("main", "B", "C", "R", "?")
        key = {n: bool((k.value >> i) & 1) for i, n in enumerate(("small", "strip", "deforming", "diffuse", "caller_stream"))}
        where = {"raster": names[(w.value >> 16) & 15], "gen": names[w.value & 15], "trace": names[(w.value >> 4) & 15], "shade": names[(w.value >> 8) & 15], "frames_in_flight": (w.value >> 12) & 15}
        return key, where

    def collapse_weights(self, area=None, tris=None):
        """(area weight, triangle-count weight) of the 4-wide collapse's objective; given both, sets them for builds from now on (diagnostic)."""
        self.L.rtggx_debug_collapse_weights.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        get = (C.c_float * 2)()
        if area is not None:
            st = (C.c_float * 2)(float(area), float(tris))
            self._check(self.L.rtggx_debug_collapse_weights(self.h, st, get))
        else:
            self._check(self.L.rtggx_debug_collapse_weights(self.h, None, get))
        return float(get[0]), float(get[1])

    def set_history_peers(self, bounds, tss0, tss1):
        """Multi-GPU strips: every rank's TemporalSSOut[0] / [1] as device pointers valid in this process (0: this rank's own), and the
        strip boundaries (world + 1 rows) -- history taps beyond the exchanged apron then read the owner's image.  bounds=None: forget them."""
        self.L.rtggx_set_history_peers.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        if bounds is None:
            self._check(self.L.rtggx_set_history_peers(self.h, 0, None, None, None))
            return
        world = len(bounds) - 1
        b = (C.c_uint32 * (world + 1))(*[int(x) for x in bounds])
        p0 = (C.c_void_p * world)(*[C.c_void_p(int(x) or None) for x in tss0])
        p1 = (C.c_void_p * world)(*[C.c_void_p(int(x) or None) for x in tss1])
        self._check(self.L.rtggx_set_history_peers(self.h, world, b, p0, p1))

    def history_ipc_export(self):
        """This context's two history images as inter-process handles (2 x 64 bytes) for another process's history_ipc_open."""
        self.L.rtggx_history_ipc_export.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        buf = C.create_string_buffer(2 * IPC_HANDLE_BYTES)
        self._check(self.L.rtggx_history_ipc_export(self.h, buf, len(buf.raw)))
        return bytes(buf.raw)

    def history_ipc_open(self, handles):
        """Another process's history_ipc_export opened here: (TemporalSSOut[0], TemporalSSOut[1]) device pointers for set_history_peers."""
        self.L.rtggx_history_ipc_open.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p]
        p0, p1 = C.c_void_p(), C.c_void_p()
        self._check(self.L.rtggx_history_ipc_open(self.h, bytes(handles), len(handles), C.byref(p0), C.byref(p1)))
        return int(p0.value), int(p1.value)

    def set_sampler(self, vndf):
        self._check(self.L.rtggx_set_sampler(self.h, 1 if vndf else 0))

    def set_refit_policy(self, rebuild_ratio=1.2, steps_per_frame=16):
        self._check(self.L.rtggx_set_refit_policy(self.h, rebuild_ratio, steps_per_frame))

    def refit_stats(self, slot=1):
        ratio, refits, rebuilds = C.c_float(), C.c_uint32(), C.c_uint32()
        self._check(self.L.rtggx_refit_stats(self.h, slot, C.byref(ratio), C.byref(refits), C.byref(rebuilds)))
        return {"cost_ratio": ratio.value, "refits": refits.value, "rebuilds": rebuilds.value}

    def set_env(self, fmt, size, mips, data):
        b = np.frombuffer(bytes(data), np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data).view(np.uint8).reshape(-1)
        self._check(self.L.rtggx_set_env(self.h, fmt, size, mips, _p(b), b.size))

    def set_material(self, mesh, base_color, roughness, metallic):
        bc = np.asarray(base_color, np.float32)
        self._check(self.L.rtggx_set_material(self.h, mesh, _p(bc), roughness, metallic))

    def set_metallic(self, mesh, metallic):
        self._check(self.L.rtggx_set_metallic(self.h, mesh, metallic))

    def build_as(self):
        self._check(self.L.rtggx_build_as(self.h))

    def update_frame(self, constants768):
        b = np.frombuffer(bytes(constants768), np.uint8).copy()
        if b.size != 768:
            raise ValueError("RtggxFrameConstants is 768 bytes")
        self._check(self.L.rtggx_update_frame(self.h, _p(b)))

    def update_as(self):
        self._check(self.L.rtggx_update_as(self.h))

    def transform_sh(self):
        self._check(self.L.rtggx_transform_sh(self.h))

    def render_visibility(self):
        self._check(self.L.rtggx_render_visibility(self.h))

    def ray_trace(self):
        self._check(self.L.rtggx_ray_trace(self.h))

    def denoise(self, use_shared_mem=False):
        self._check(self.L.rtggx_denoise(self.h, 1 if use_shared_mem else 0))

    def tone_map(self):
        self._check(self.L.rtggx_tone_map(self.h))

    def sync(self):
        self._check(self.L.rtggx_sync(self.h))

    def ray_count(self):
        n = C.c_uint64()
        self._check(self.L.rtggx_ray_count(self.h, C.byref(n)))
        return int(n.value)

    def ray_total(self, reset=False):
        n = C.c_uint64()
        self._check(self.L.rtggx_ray_total(self.h, C.byref(n), 1 if reset else 0))
        return int(n.value)

    def debug_trace_split(self, work_per_wave, max_shift, capacity=-1):
        """Sets the adaptive-split parameters of the trace kernel; returns the split-list entries the last frame asked for."""
        d = C.c_uint32()
        self.L.rtggx_debug_trace_split.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
        self._check(self.L.rtggx_debug_trace_split(self.h, work_per_wave, max_shift, capacity, C.byref(d)))
        return int(d.value)

    def stream(self):
        """The context's main stream as an integer handle (rtggx_get_stream)."""
        h = C.c_void_p()
        self.L.rtggx_get_stream.argtypes = [C.c_void_p, C.c_void_p]
        self._check(self.L.rtggx_get_stream(self.h, C.byref(h)))
        return int(h.value or 0)

    def trace_residency(self, force_waves=0):
        """(waves of the traversal's resident workgroup, share of the frame period the traversal took when last sampled)."""
        w, sh = C.c_uint32(), C.c_float()
        self.L.rtggx_debug_trace_residency.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        self._check(self.L.rtggx_debug_trace_residency(self.h, force_waves, C.byref(w), C.byref(sh)))
        return int(w.value), float(sh.value)

    def debug_counters(self, n=8, reset=True):
        out = np.zeros(n, np.uint32)
        self.L.rtggx_debug_counters.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]
        self._check(self.L.rtggx_debug_counters(self.h, _p(out), n, 1 if reset else 0))
        return out

    def enable_timing(self, mode=1):
        """0 off, 1 every pass (timings()), 2 ray-trace kernel ring (kernel_times()), 3 the same for every 8th frame."""
        self._check(self.L.rtggx_enable_timing(self.h, int(mode)))

    def kernel_times(self):
        ms = np.zeros(4096, np.float32)
        n = C.c_uint32()
        self._check(self.L.rtggx_kernel_times(self.h, _p(ms), ms.size, C.byref(n)))
        return ms[:n.value].copy()

    def timings(self):
        t = Timings()
        self._check(self.L.rtggx_get_timings(self.h, C.byref(t)))
        return {n: getattr(t, n) for n, _ in Timings._fields_}

    def buffer_size(self, bid):
        n = C.c_size_t()
        self._check(self.L.rtggx_buffer_size(self.h, bid, C.byref(n)))
        return int(n.value)

    def buffer_ptr(self, bid):
        p = C.c_void_p()
        self._check(self.L.rtggx_buffer_ptr(self.h, bid, C.byref(p)))
        return int(p.value)

    def readback(self, bid):
        n = self.buffer_size(bid)
        dt = np.dtype(_BUF_DTYPE[bid])
        out = np.zeros(max(n // dt.itemsize, 0), dt)
        if n:
            self._check(self.L.rtggx_readback(self.h, bid, _p(out), n))
        if bid <= BUF_BACKBUFFER:
            return out.reshape(self.H, self.W)
        if bid == BUF_SH_COEFFS:
            return out.reshape(9, 3)
        if bid in (BUF_BVH_NODES0, BUF_BVH_NODES1):
            return out.reshape(-1, 16)
        if bid in (BUF_BVH_TRIS0, BUF_BVH_TRIS1):
            return out.reshape(-1, 16)
        if bid == BUF_TLAS:
            return out.reshape(2, 4, 4)
        if bid == BUF_ENV:
            return out.reshape(-1, 4)
        if bid in (BUF_BVH4_NODES0, BUF_BVH4_NODES1, BUF_BVH4_TOP0, BUF_BVH4_TOP1):
            return out.reshape(-1, 32)
        return out

    def upload(self, bid, array):
        a = np.ascontiguousarray(array)
        self._check(self.L.rtggx_upload(self.h, bid, _p(a), a.nbytes))

    def frame_parity(self):
        p = C.c_uint32()
        self._check(self.L.rtggx_frame_parity(self.h, C.byref(p)))
        return int(p.value)

    def bvh_root(self, slot):
        r = C.c_int32()
        self._check(self.L.rtggx_bvh_root(self.h, slot, C.byref(r)))
        return int(r.value)

    def trace_rays(self, rays):
        r = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        out = np.zeros((r.shape[0], 6), np.float32)
        self._check(self.L.rtggx_trace_rays(self.h, _p(r), r.shape[0], _p(out)))
        return {"t": out[:, 0].copy(), "inst": out[:, 1].copy().view(np.uint32), "prim": out[:, 2].copy().view(np.uint32),
                "b1": out[:, 3].copy(), "b2": out[:, 4].copy(), "valid": out[:, 5] > 0.5}
