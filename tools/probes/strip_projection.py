"""What an N-GPU frame would take, projected from ONE GPU: per N and strip, the strip rendered alone with balanced boundaries --
   compute    render only, no exchange
   +rccl      render + the rank's OWN exchange plan issued every frame through the real RCCL group on the rendering stream, against a
              one-rank communicator: every send of the plan is paired with a receive into a scratch buffer and every receive with a
              send from one, so the group launch, its kernel and its bytes are all there -- only the wire is missing.  Round 4: the plan
              holds the ordering tokens between ranks that exchange nothing else (strips.exchange_plan; rtggx_set_history_peers)
   no tokens  the same without them (rounds 2-3's plan): what the tokens cost
   wire       the bytes the rank sends / receives per frame and what they take over xGMI at 153 GB/s per link (history rows go to the
              two neighbours, one link each; rank 0 receives N - 1 back-buffer strips over N - 1 links in parallel): not included in
              +rccl (an intra-GPU copy stands in), listed beside it
and the slowest strip of each, rank 0 (which gathers the frame) separately.   python tools/probes/strip_projection.py [W H]
(PROJ_NS=8 PROJ_MODES=compute,+rccl for a subset; RTGGX_GATHER_ROW_WEIGHT: strips.py, what a row rank 0 receives weighs in the balance)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import assets
from raytracedggx_amd import rccl
from raytracedggx_amd.strips import StripRenderer, exchange_plan
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
FRAMES = 300
LINK_GBS = 153.0
mesh, env = assets.path("bunny.obj"), assets.path("rnl_cross.dds")
comm = rccl.Communicator(None, 0, 1)
scratch = torch.empty(W * H * 8, dtype=torch.uint8, device="cuda")      # stands in for the peers' buffers
print("%dx%d, bunny, %d frames per measurement; ms per frame" % (W, H, FRAMES), flush=True)
base = None
NS = tuple(int(x) for x in os.environ.get("PROJ_NS", "1,2,4,8").split(","))                    # e.g. PROJ_NS=8 PROJ_MODES=compute for an A/B
MODES = tuple(os.environ.get("PROJ_MODES", "compute,+rccl,no tokens").split(","))
for N in NS:
    bounds = None
    if N > 1:
        p = StripRenderer(W, H, mesh, env, rank=0, world=N, transport=lambda *_: None, extra_args=("-sharedmem",), balance=True)
        bounds = p.bounds; p.close()
    rows = []
    for r in range(N):
        plans = {"+rccl": exchange_plan(H, r, N, bounds=bounds) if N > 1 else [], "no tokens": exchange_plan(H, r, N, bounds=bounds, tokens=False) if N > 1 else []}
        size = lambda name, r0, r1: (r1 - r0) * (4 if name == "token" else W * (8 if name == "history" else 4))
        sent = sum(size(name, r0, r1) for op, name, r0, r1, peer in plans["+rccl"] if op == "send")
        recv = sum(size(name, r0, r1) for op, name, r0, r1, peer in plans["+rccl"] if op == "recv")
        # over the wire: every transfer has a link of its own except the back-buffer strips rank 0 receives (one link per sender, in parallel)
        wire_us = max([size(name, r0, r1) / (LINK_GBS * 1e3) for op, name, r0, r1, peer in plans["+rccl"]] or [0.0])
        res = {m: float("nan") for m in ("compute", "+rccl", "no tokens")}
        for mode in MODES:
            def self_exchange(renderer, plan_, mode=mode):
                ops, off = [], 0
                for is_send, ptr, nbytes, peer in renderer.raw_ops(plans[mode], renderer.context.frame_parity()):
                    ops.append((is_send, ptr, nbytes, 0))
                    ops.append((not is_send, scratch.data_ptr() + off, nbytes, 0))
                    off += (nbytes + 255) // 256 * 256
                comm.exchange(ops, renderer.stream.cuda_stream)
            transport = (lambda *_: None) if (mode == "compute" or N == 1) else self_exchange
            s = StripRenderer(W, H, mesh, env, rank=r, world=N, transport=transport if N > 1 else None, torch_buffers=N > 1, extra_args=("-sharedmem",), balance=bounds if N > 1 else False)
            for _ in range(FRAMES): s.frame()
            s.context.sync(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(FRAMES): s.frame()
            s.context.sync(); torch.cuda.synchronize()
            res[mode] = (time.perf_counter() - t0) / FRAMES * 1e3
            s.close()
        rows.append((r, res["compute"], res["+rccl"], sent, recv, wire_us, res["no tokens"]))
        print("  N=%d strip %d: compute %.4f  +rccl %.4f  (without the tokens %.4f)   %d ops, sends %7.0f KB, receives %7.0f KB per frame, longest transfer on the wire %.1f us" % (
            N, r, res["compute"], res["+rccl"], res["no tokens"], len(plans["+rccl"]), sent / 1e3, recv / 1e3, wire_us), flush=True)
    slow_c, slow_x, slow_b = max(x[1] for x in rows), max(x[2] for x in rows), max(x[6] for x in rows)
    if N == 1: base = slow_c
    print("N=%d: slowest strip compute %.4f ms (x%.2f), with its exchange through the RCCL group %.4f ms (x%.2f; rank 0, which gathers: %.4f; slowest other: %.4f), without the tokens %.4f ms; rank 0 gathers %.0f KB per frame (%.1f us over %d links in parallel); bounds %s" % (
        N, slow_c, base / slow_c, slow_x, base / slow_x, rows[0][2], max([x[2] for x in rows[1:]] or [float("nan")]), slow_b, rows[0][4] / 1e3,
        max([0.0] + [(bounds[k + 1] - bounds[k]) * W * 4 / (LINK_GBS * 1e3) for k in range(1, N)]) if N > 1 else 0.0, max(N - 1, 0), bounds), flush=True)
comm.destroy()
