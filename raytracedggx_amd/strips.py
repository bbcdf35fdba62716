"""Screen-space row strips across the GPUs of one node, one process per GPU.

Rank r of N owns rows [r*H/N, (r+1)*H/N) of the frame.  Every pass that is a pure function of the pixel
(visibility, ray trace, the horizontal filters) simply recomputes the apron rows it needs
(rtggx_set_strip: +-18 rows for the G-buffer passes, +-2 for the vertical filters, +-1 for the temporal
pass); nothing is exchanged for the spatial passes.  The ONE real exchange step is the temporal history:
TemporalSSOut rows next to the strip boundary are produced by the neighbouring rank, and the next frame's
reprojection reads them.  After each frame every rank sends its HISTORY_APRON boundary rows of
TemporalSSOut[parity] to its two neighbours and receives theirs -- point-to-point over xGMI
(RCCL send/recv, one group launch per frame, 2 x 17 rows x W x 8 B), together with the gather of the
tone-mapped strips onto rank 0.  The exchange is issued right after the tone map and is only needed by
the next frame's temporal pass, so it overlaps that frame's visibility, ray trace and spatial passes.

With world == 1 this is exactly RayTracedGGX::OnUpdate + OnRender and no communication.
"""
import numpy as np

from . import app, capi

HISTORY_APRON = 17   # rows: bilinear footprint (1) + the largest vertical reprojection distance handled exactly (16 px/frame)


class _DeviceArray:
    """Exposes a raw device pointer through __cuda_array_interface__ so torch can wrap it without a copy."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 3}


def strip_rows(height, rank, world):
    return (rank * height) // world, ((rank + 1) * height) // world


class StripRenderer:
    def __init__(self, width, height, mesh_path, env_path, rank=0, world=1, device=0, dist=None, pos_scale=None, extra_args=()):
        self.W, self.H, self.rank, self.world, self.dist = width, height, rank, world, dist
        args = ["-mesh", mesh_path] + ([str(x) for x in pos_scale] if pos_scale else []) + \
               ["-env", env_path, "-width", width, "-height", height, "-device", device] + list(extra_args)
        self.app = app.RayTracedGGX(args)
        self.context = self.app.context
        self.b, self.e = strip_rows(height, rank, world)
        if world > 1:
            if self.e - self.b < HISTORY_APRON:
                raise ValueError("strips of %d rows are thinner than the %d-row history apron" % (self.e - self.b, HISTORY_APRON))
            import torch
            self.torch = torch
            self.context.set_strip(self.b, self.e)
            # run the HIP passes on torch's current stream so that RCCL ops and kernels are ordered by the stream
            self.context.set_stream(torch.cuda.current_stream().cuda_stream)
            self._tss = [self._wrap(capi.BUF_TSS0, "<u8"), self._wrap(capi.BUF_TSS1, "<u8")]
            self._backbuffer = self._wrap(capi.BUF_BACKBUFFER, "<u4")
        self._last = None

    def _wrap(self, bid, typestr):
        t = self.torch.as_tensor(_DeviceArray(self.context.buffer_ptr(bid), (self.H, self.W), typestr), device="cuda")
        return t.view(self.torch.int64 if typestr == "<u8" else self.torch.int32)

    # -- one frame --------------------------------------------------------------------------------------
    def frame(self):
        self.app.OnUpdate()
        self.app.OnRender()
        if self.world > 1:
            self._exchange()

    def _exchange(self):
        dist, torch = self.dist, self.torch
        A, b, e = HISTORY_APRON, self.b, self.e
        tss = self._tss[self.context.frame_parity()]
        ops = []
        if self.rank > 0:                       # upper neighbour owns [.., b)
            ops.append(dist.P2POp(dist.isend, tss[b:b + A], self.rank - 1))
            ops.append(dist.P2POp(dist.irecv, tss[b - A:b], self.rank - 1))
        if self.rank < self.world - 1:          # lower neighbour owns [e, ..)
            ops.append(dist.P2POp(dist.isend, tss[e - A:e], self.rank + 1))
            ops.append(dist.P2POp(dist.irecv, tss[e:e + A], self.rank + 1))
        # frame assembly on rank 0 (the reference presents one back buffer)
        if self.rank == 0:
            for r in range(1, self.world):
                rb, re = strip_rows(self.H, r, self.world)
                ops.append(dist.P2POp(dist.irecv, self._backbuffer[rb:re], r))
        else:
            ops.append(dist.P2POp(dist.isend, self._backbuffer[b:e], 0))
        for w in dist.batch_isend_irecv(ops):
            w.wait()                            # stream-side wait only: the host does not block

    # -- statistics --------------------------------------------------------------------------------------
    def rays_traced_since_reset(self):
        return self.context.ray_total(reset=True)

    def ray_kernel_ms_since_reset(self):
        return self.context.kernel_times()

    def strip_rows_with_apron(self):
        return max(self.b - 18, 0), min(self.e + 18, self.H)

    def ray_kernel_algorithmic_bytes(self, rows):
        """DESIGN.md "Roofline": 18 B/pixel (visibility 4 in; normal 4, roughMetal 2, velocity 4, reflection 4 out;
        +4 when a diffuse ray is traced) + the scene arrays once (BVH nodes, leaf triangles, vertices, indices)."""
        px = (rows[1] - rows[0]) * self.W
        scene = 0
        for bid in (capi.BUF_BVH_NODES0, capi.BUF_BVH_TRIS0, capi.BUF_BVH_NODES1, capi.BUF_BVH_TRIS1):
            scene += self.context.buffer_size(bid)
        scene += self.scene_vertex_index_bytes()
        return 18 * px + scene

    def scene_vertex_index_bytes(self):
        # vertices 24 B, indices 4 B: from the leaf-triangle counts (3 indices per triangle); vertex counts from the importer
        nt = (self.context.buffer_size(capi.BUF_BVH_TRIS0) + self.context.buffer_size(capi.BUF_BVH_TRIS1)) // 64
        return 12 * nt + 24 * (24 + self._model_vertices())

    def _model_vertices(self):
        if not hasattr(self, "_nv"):
            self._nv = int(np.unique(self.context.readback(capi.BUF_BVH_TRIS1).view(np.float32).reshape(-1, 16)[:, :9].reshape(-1, 3), axis=0).shape[0])
        return self._nv

    def last_timings(self):
        """Per-pass milliseconds of one extra, fully instrumented frame (outside any timed region)."""
        self.context.enable_timing(1)
        self.frame()
        self.context.sync()
        t = {k: round(v, 4) for k, v in self.context.timings().items()}
        self.context.enable_timing(0)
        return t

    def close(self):
        self.app.OnDestroy()
