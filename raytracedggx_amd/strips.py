"""Screen-space row strips across the GPUs of one node, one process per GPU.

Rank r of N owns rows [r*H/N, (r+1)*H/N) of the frame.  Every pass that is a pure function of the pixel
(visibility, ray trace, the horizontal filters) simply recomputes the apron rows it needs
(rtggx_set_strip: +-18 rows for the G-buffer passes, +-2 for the vertical filters, +-1 for the temporal
pass); nothing is exchanged for the spatial passes.  The ONE real exchange step is the temporal history:
TemporalSSOut rows next to the strip boundary are produced by the neighbouring rank, and the next frame's
reprojection reads them.  After each frame every rank sends its HISTORY_APRON boundary rows of
TemporalSSOut[parity] to its two neighbours and receives theirs -- point-to-point over xGMI
(RCCL send/recv, one group launch per frame, 2 x 18 rows x W x 8 B), together with the gather of the
tone-mapped strips onto rank 0.  The exchange is issued right after the tone map and is only needed by
the next frame's temporal pass, so it overlaps that frame's visibility, ray trace and spatial passes.
torch.distributed makes the rendezvous; the per-frame sends and receives go to RCCL directly (rccl.py: the
P2POp batch of torch.distributed costs more host time per frame than a thin strip takes to render), unless
RTGGX_EXCHANGE=torch asks for the torch.distributed batch.

With world == 1 this is exactly RayTracedGGX::OnUpdate + OnRender and no communication.
"""
import os

import numpy as np

from . import app, capi

HISTORY_APRON = 18   # rows: temporal apron (1: the pass also computes rows b-1 and e for the tone map) + the largest vertical reprojection handled exactly (16 px/frame) + bilinear footprint (1)


class _DeviceArray:
    """Exposes a raw device pointer through __cuda_array_interface__ so torch can wrap it without a copy."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 3}


def strip_rows(height, rank, world, bounds=None):
    """Rows [begin, end) of `rank`: equal strips, or the boundaries `bounds` (world + 1 ascending row numbers, 0 .. height)."""
    if bounds is not None:
        return int(bounds[rank]), int(bounds[rank + 1])
    return (rank * height) // world, ((rank + 1) * height) // world


def balanced_bounds(row_cost, world, min_rows=HISTORY_APRON, first_extra=0.0):
    """Strip boundaries that even out sum(row_cost) per strip: boundary k at the row where the running cost passes k/world of
    the total, every strip at least `min_rows` rows (the history apron must fit).  first_extra: a cost rank 0 carries besides its
    rows -- it receives everybody's back-buffer strip (gather_cost) -- so its strip ends where its rows cost that much LESS than the
    others'.  Deterministic in its inputs: every rank computes the same boundaries from the same profile."""
    cost = np.asarray(row_cost, np.float64)
    height = len(cost)
    if height < world * min_rows:
        raise ValueError("%d rows cannot hold %d strips of at least %d rows" % (height, world, min_rows))
    csum = np.concatenate([[0.0], np.cumsum(cost)])
    share = (csum[-1] + first_extra) / world
    bounds = [0]
    for k in range(1, world):
        b = int(np.searchsorted(csum, share * k - first_extra, side="left"))
        b = max(b, bounds[-1] + min_rows)                       # room for this strip ...
        b = min(b, height - (world - k) * min_rows)             # ... and for the ones that follow
        bounds.append(b)
    bounds.append(height)
    return bounds


def gather_cost(width, height, world, row_weight):
    """What rank 0's gather of the frame weighs in the balance, in the row costs' unit (covered pixels): row_weight x width per row it
    receives, (world - 1) / world of the frame (StripRenderer.GATHER_ROW_WEIGHT; measured: tools/probes/strip_projection.py)."""
    return row_weight * width * height * (world - 1) / world if world > 1 else 0.0


def exchange_plan(height, rank, world, apron=HISTORY_APRON, bounds=None, tokens=True):
    """The point-to-point transfers of one frame for `rank`: a list of (op, buffer, row_begin, row_end, peer) with op in
    {"send", "recv"} and buffer in {"history", "backbuffer", "token"}.  Rows are frame rows: every rank allocates full-size
    targets, so a transfer reads and writes the same rows on both sides.  Ops between a pair of ranks appear in the same
    order on both (history first, then the back-buffer strip, then the token), which is what tag-less send/recv matching needs.
    tokens (round 4): with every rank's history image mapped into every rank (rtggx_set_history_peers) a history tap beyond the apron
    reads the owner's image, and that needs an order between ANY two ranks, in both directions, every frame: the reader's temporal pass
    behind the owner's previous one, the owner's next-but-one H filter (which reuses the image as scratch) behind the reader's.  A
    message each way in this exchange gives exactly that (it is issued on the main streams between the two).  Neighbours have their
    history rows, everybody has a back-buffer strip for rank 0; the pairs and directions that have nothing get a 4-byte token
    ("token" rows are WORDS of the context's token buffer: [rank] to send from, [MAX_PEERS + peer] to receive into)."""
    b, e = strip_rows(height, rank, world, bounds)
    ops = []
    if rank > 0:                       # upper neighbour owns [.., b)
        ops.append(("send", "history", b, min(b + apron, e), rank - 1))
        ops.append(("recv", "history", max(b - apron, 0), b, rank - 1))
    if rank < world - 1:               # lower neighbour owns [e, ..)
        ops.append(("send", "history", max(e - apron, b), e, rank + 1))
        ops.append(("recv", "history", e, min(e + apron, height), rank + 1))
    if rank == 0:                      # frame assembly on rank 0 (the reference presents one back buffer)
        for r in range(1, world):
            rb, re = strip_rows(height, r, world, bounds)
            ops.append(("recv", "backbuffer", rb, re, r))
    else:
        ops.append(("send", "backbuffer", b, e, 0))
    if tokens:
        if world > capi.MAX_PEERS:
            raise ValueError("at most %d ranks can map each other's history images" % capi.MAX_PEERS)
        for p in range(world):
            if abs(p - rank) < 2:
                continue               # myself, or a neighbour: the history rows go both ways
            if p != 0:                 # (to rank 0 goes my back-buffer strip)
                ops.append(("send", "token", rank, rank + 1, p))
            if rank != 0:              # (rank 0 receives p's strip)
                ops.append(("recv", "token", capi.MAX_PEERS + p, capi.MAX_PEERS + p + 1, p))
    return ops


def plan_to_raw(plan, history_ptr, backbuffer_ptr, width, token_ptr=0):
    """`plan` as the argument lists of ncclSend / ncclRecv: (is_send, device pointer, bytes, peer).  Rows [r0, r1) of
    TemporalSSOut[parity] (8 B/px) or of the back buffer (4 B/px); every rank holds full-size targets, so both sides
    address the same rows.  Tokens: words [r0, r1) of the context's token buffer."""
    base = {"history": (history_ptr, 8 * width), "backbuffer": (backbuffer_ptr, 4 * width), "token": (token_ptr, 4)}
    return [(op == "send", base[name][0] + r0 * base[name][1], (r1 - r0) * base[name][1], peer) for op, name, r0, r1, peer in plan]


def make_ops(dist, plan, buffers):
    """The P2P operations of `plan` over `buffers` = {"history": tensor[H, W], "backbuffer": tensor[H, W]} (row slices
    are views: the list can be built once and reused every frame)."""
    return [dist.P2POp(dist.isend if op == "send" else dist.irecv, buffers[name][r0:r1], peer) for op, name, r0, r1, peer in plan]


def run_exchange(dist, plan, buffers, ops=None):
    """Issue `plan` over torch.distributed as one batch (RCCL group launch on GPU tensors; plain isend/irecv on gloo)."""
    ops = make_ops(dist, plan, buffers) if ops is None else ops
    for w in dist.batch_isend_irecv(ops) if ops else []:
        w.wait()                       # on GPU tensors a stream-side wait only: the host does not block


class StripRenderer:
    PROFILE_FRAMES = 2         # full frames every rank renders first when it balances the strips itself
    # cost of a row = covered pixels + this x width (rows without a surface are not free).  0.3 until the traversal got cheaper relative to
    # the per-pixel passes; 0.5 now: slowest of 8 strips at 4K 0.158 -> 0.145 ms, of 4 at 1080p 0.089 -> 0.085 (profiles/r02_n_strip_projection.txt)
    SKY_ROW_WEIGHT = float(os.environ.get("RTGGX_SKY_ROW_WEIGHT", 0.5))
    # what a row rank 0 RECEIVES (the gather of the tone-mapped strips) costs it, in the same unit (x width): round 3's projection had rank 0
    # as the slowest of 8 strips because it gathers (compute 0.058 ms, with its exchange 0.100; profiles/r03_h_strip_projection.txt)
    GATHER_ROW_WEIGHT = float(os.environ.get("RTGGX_GATHER_ROW_WEIGHT", 0.02))

    def __init__(self, width, height, mesh_path, env_path, rank=0, world=1, device=0, dist=None, pos_scale=None, extra_args=(),
                 transport=None, torch_buffers=None, balance=False, apron=HISTORY_APRON, peers=True):
        """dist: torch.distributed (one process per GPU).  transport: instead of dist, a callable
        transport(renderer, plan) that carries out the plan some other way (tests drive several strips from one process).
        torch_buffers: wrap the exchanged targets as torch tensors and render on torch's current stream (default: only
        when dist is used); with a transport it lets a test move the rows with torch copies on that stream.
        balance: False = equal strips; a list of world + 1 row numbers = these boundaries; True = every rank first renders
        PROFILE_FRAMES full frames and cuts the frame where the covered pixels (= rays, the expensive rows) balance --
        rendering is deterministic, so all ranks arrive at the same boundaries without talking to each other.
        peers: map every rank's history images into every rank (rtggx_set_history_peers) so that a history tap beyond the exchanged apron
        reads the owner's image -- N strips then equal the single-GPU frame at any velocity -- and add the ordering tokens to the exchange
        (exchange_plan).  With dist the handles travel through an all_gather (hipIpc); several strips in ONE process (a transport) are
        connected by connect_peers(all_of_them) once they exist.
        apron: history rows exchanged beyond each strip edge (HISTORY_APRON = 18 covers 16 px of vertical reprojection per
        frame).  A faster motion -- an orbit drag of a -track script -- makes the strips differ from the single-GPU frame:
        the temporal pass detects that (history_overreach() > 0), and a wider apron, the same on every rank, is the remedy."""
        self.W, self.H, self.rank, self.world, self.dist, self.transport = width, height, rank, world, dist, transport
        self.apron = int(apron)
        self.peers = bool(peers) and world > 1
        args = ["-mesh", mesh_path] + ([str(x) for x in pos_scale] if pos_scale else []) + \
               ["-env", env_path, "-width", width, "-height", height, "-device", device] + list(extra_args)
        self.app = app.RayTracedGGX(args)
        self.context = self.app.context
        self.bounds = None
        if world > 1 and balance is not False and balance is not None:
            self.bounds = self.profile_bounds(world) if balance is True else [int(x) for x in balance]
            if len(self.bounds) != world + 1 or self.bounds[0] != 0 or self.bounds[-1] != height:
                raise ValueError("balance: %d boundaries from 0 to %d expected, got %s" % (world + 1, height, self.bounds))
        self.b, self.e = strip_rows(height, rank, world, self.bounds)
        if world > 1:
            if self.e - self.b < self.apron:
                raise ValueError("strips of %d rows are thinner than the %d-row history apron" % (self.e - self.b, self.apron))
            self.context.set_strip(self.b, self.e)
            self.context.set_history_apron(self.apron)
            if transport is None or torch_buffers:
                import torch
                self.torch = torch
                # The exchange is ordered behind the frame on the context's OWN main stream, wrapped for torch (ExternalStream) -- not on a
                # stream torch creates: the library runs a frame on four streams, as many as HIP gives hardware queues by default, and a
                # fifth stream shares a queue with one of them and serialises with it (round 3: a 1080p strip of eight took 0.124 ms on a
                # torch stream against 0.076 on the library's own; profiles/r03_h_strip_projection.txt).  Asking for the stream also tells
                # the library that the caller orders work of its own behind the frame (the tone map stays on this stream).
                self.stream = torch.cuda.ExternalStream(self.context.stream(), device=device)
                self.xstream = self.stream      # (rounds 2-3 had an exchange stream of its own here: measured slower, removed in round 4)
                self._tss = [self._wrap(capi.BUF_TSS0, "<u8"), self._wrap(capi.BUF_TSS1, "<u8")]
                self._backbuffer = self._wrap(capi.BUF_BACKBUFFER, "<u4")
        self._last = None
        self._ops = [None, None]
        self._comm = None
        if world > 1 and dist is not None and transport is None and os.environ.get("RTGGX_EXCHANGE", "rccl") != "torch":
            from . import rccl
            try:
                self._comm = rccl.Communicator(dist, rank, world)
            except RuntimeError as e:          # raised on every rank or on none (rccl.Communicator agrees on it first)
                if rank == 0:
                    import sys
                    print("strips: %s -- using torch.distributed's P2P batch instead" % e, file=sys.stderr, flush=True)
                self._comm = None
        if self.peers and dist is not None and transport is None:
            self._connect_peers_over(dist)

    def plan(self):
        return exchange_plan(self.H, self.rank, self.world, apron=self.apron, bounds=self.bounds, tokens=self.peers)

    def all_bounds(self):
        return self.bounds if self.bounds is not None else [strip_rows(self.H, r, self.world)[0] for r in range(self.world)] + [self.H]

    def connect_peers(self, renderers):
        """Several strips in one process: every strip gets the history images of all of them as plain device pointers."""
        if not self.peers:
            return
        self.context.set_history_peers(self.all_bounds(), [r.context.buffer_ptr(capi.BUF_TSS0) for r in renderers], [r.context.buffer_ptr(capi.BUF_TSS1) for r in renderers])

    def _connect_peers_over(self, dist):
        """One process per GPU: the handles of every rank's two history images, all-gathered, opened here (hipIpcOpenMemHandle)."""
        import torch
        mine = torch.frombuffer(bytearray(self.context.history_ipc_export()), dtype=torch.uint8)
        on_gpu = dist.get_backend() != "gloo"
        if on_gpu:
            mine = mine.cuda()
        every = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(every, mine)
        p0, p1 = [], []
        for r in range(self.world):
            if r == self.rank:
                p0.append(0); p1.append(0)
            else:
                a, b = self.context.history_ipc_open(every[r].cpu().numpy().tobytes())
                p0.append(a); p1.append(b)
        self.context.set_history_peers(self.all_bounds(), p0, p1)
        dist.barrier()      # nobody renders into an image another rank has not finished mapping

    def _wrap(self, bid, typestr):
        t = self.torch.as_tensor(_DeviceArray(self.context.buffer_ptr(bid), (self.H, self.W), typestr), device="cuda")
        return t.view(self.torch.int64 if typestr == "<u8" else self.torch.int32)

    def profile_bounds(self, world):
        """Renders PROFILE_FRAMES whole frames and returns the boundaries that balance covered pixels (+ a per-row constant)."""
        for _ in range(self.PROFILE_FRAMES):
            self.render()
        self.context.sync()
        covered = (self.context.readback(capi.BUF_VISIBILITY) != 0).sum(axis=1)
        return balanced_bounds(covered + self.SKY_ROW_WEIGHT * self.W, world, min_rows=self.apron, first_extra=gather_cost(self.W, self.H, world, self.GATHER_ROW_WEIGHT))

    # -- one frame --------------------------------------------------------------------------------------
    def frame(self):
        self.render()
        self.exchange()

    def render(self):
        self.app.OnUpdate()
        self.app.OnRender()

    def exchange(self):
        if self.world == 1:
            return
        stream = getattr(self, "stream", None)
        if self.transport is not None:
            self.transport(self, self.plan())
            return
        parity = self.context.frame_parity()
        if self._comm is not None:
            if self._ops[parity] is None:      # (is_send, pointer, bytes, peer), built once per history target
                self._ops[parity] = self.raw_ops(self.plan(), parity)
            self._comm.exchange(self._ops[parity], stream.cuda_stream)
            return
        if self._ops[parity] is None:          # built once per history target: the per-frame host cost is the batch call alone
            self._ops[parity] = make_ops(self.dist, self.plan(), self.exchange_buffers())
        with self.torch.cuda.stream(stream):
            run_exchange(self.dist, None, None, ops=self._ops[parity])

    def raw_ops(self, plan, parity):
        """`plan` for rccl.Communicator.exchange, over this renderer's TemporalSSOut[parity] and back buffer."""
        return plan_to_raw(plan, self.context.buffer_ptr(capi.BUF_TSS1 if parity else capi.BUF_TSS0), self.context.buffer_ptr(capi.BUF_BACKBUFFER), self.W,
                           self.context.buffer_ptr(capi.BUF_EXCHANGE_TOKENS))

    def exchange_buffers(self):
        """The torch views of the two exchanged targets (this frame's temporal result, the back buffer)."""
        if not hasattr(self, "_tokens"):
            self._tokens = self.torch.as_tensor(_DeviceArray(self.context.buffer_ptr(capi.BUF_EXCHANGE_TOKENS), (2 * capi.MAX_PEERS,), "<u4"), device="cuda").view(self.torch.int32)
        return {"history": self._tss[self.context.frame_parity()], "backbuffer": self._backbuffer, "token": self._tokens}

    def history_overreach(self, reset=True):
        """Rows by which this rank's temporal pass read history beyond the exchanged apron since the last reset (synchronises).
        0: every frame so far equals the single-GPU frame; otherwise the motion was faster than `apron` covers."""
        return self.context.history_overreach(reset) if self.world > 1 else 0

    # -- statistics --------------------------------------------------------------------------------------
    def rays_traced_since_reset(self):
        return self.context.ray_total(reset=True)

    def ray_kernel_ms_since_reset(self):
        return self.context.kernel_times()

    def strip_rows_with_apron(self):
        return max(self.b - 18, 0), min(self.e + 18, self.H)

    def trace_kernel_algorithmic_bytes(self, rays_per_launch, per_ray=40):
        """DESIGN.md "Roofline": what one launch of traceKernel has to move if every byte moved once: per ray the 32 bytes of the
        48-byte record the traversal reads (origin, direction, pixel, start primitive: csrc/rt_queue.h) and the 8-byte hit key it
        merges = 40 B, plus the acceleration structure once (the 4-wide nodes in use and the leaf triangles of both instances).
        per_ray=72 gives round 1's figure (64-byte record + 8-byte key), kept in the bench line as `algorithmic_bytes_r01`."""
        return per_ray * rays_per_launch + self.bvh_bytes()

    def bvh_bytes(self):
        if not hasattr(self, "_bvh_bytes"):
            n = 0
            for b4, bt in ((capi.BUF_BVH4_NODES0, capi.BUF_BVH_TRIS0), (capi.BUF_BVH4_NODES1, capi.BUF_BVH_TRIS1)):
                nodes4 = self.context.readback(b4)
                n += 128 * int(nodes4.any(axis=1).sum()) + self.context.buffer_size(bt)
            self._bvh_bytes = n
        return self._bvh_bytes

    def frame_algorithmic_bytes(self, rows, metallic_lt_1=False, survey=False):
        """Bytes of one frame if every pass read each input and wrote each output once (SURVEY.md 8(d) per pass) + the scene once.
        survey=True: SURVEY's sum, 146 B/pixel for the all-metal scene (178 with diffuse rays).  Default: only the passes this
        build LAUNCHES -- on all-metal frames the two diffuse filter passes (6 + 22 B/pixel) are not launched, 118 B/pixel."""
        per_px = 178 if metallic_lt_1 else (146 if survey else 118)
        return per_px * (rows[1] - rows[0]) * self.W + self.bvh_bytes()

    def last_timings(self):
        """Per-pass milliseconds of one extra, fully instrumented frame (outside any timed region)."""
        self.context.enable_timing(1)
        self.render()                    # no exchange: this may be called by one rank alone
        self.context.sync()
        t = {k: round(v, 4) for k, v in self.context.timings().items() if k != "update_as"}     # (update_as: a 912-byte upload, its "duration" is queueing behind the previous frame)
        self.context.enable_timing(0)
        return t

    def close(self):
        if self._comm is not None:
            self.context.sync()
            self._comm.destroy()
            self._comm = None
        self.app.OnDestroy()
