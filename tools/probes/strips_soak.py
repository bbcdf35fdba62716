"""The 8-rank shape on one GPU for many frames: 8 balanced strips exchanging through the real RCCL group, the assembled back buffer compared
with the single-context frame EVERY frame (the GPU tests do 2-6 frames).  Round 4: every strip maps every strip's history images (rtggx_set_history_peers), a history tap
beyond the 18 exchanged rows reads the owner's image, and NO frame may differ; the apron guard (rtggx_history_overreach) still counts such
taps.  With `nopeers` (rounds 2-3) a frame may differ from the first report on.   python tools/probes/strips_soak.py [frames] [nopeers]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import assets
from raytracedggx_amd import capi, rccl
from raytracedggx_amd.strips import StripRenderer, HISTORY_APRON
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 300
peers = not (len(sys.argv) > 2 and sys.argv[2] == "nopeers")      # "nopeers": rounds 2-3's behaviour (a tap beyond the apron is reported, not served)
world = 8
mesh, env = assets.path("bunny.obj"), assets.path("rnl_cross.dds")
errors = 0
for W, H, extra in ((480, 272, ()), (1920, 1080, ()), (1280, 720, ("-metallic", 0.25, 0.5))):
    comm = rccl.Communicator(None, 0, 1)
    strips = []
    def transport(r, plan):      # as tests/test_gpu_parity.py _strips_through_rccl_equal_the_full_frame
        ops = []
        for op, name, r0, r1, peer in plan:
            if op == "recv":
                src = strips[peer]
                ops += src.raw_ops([("send", name, r0, r1, 0)], src.context.frame_parity())
                ops += r.raw_ops([("recv", name, r0, r1, 0)], r.context.frame_parity())
        for t in strips:
            r.xstream.wait_stream(t.xstream); r.xstream.wait_stream(t.stream)
        comm.exchange(ops, r.xstream.cuda_stream)
    full = StripRenderer(W, H, mesh, env, extra_args=("-sharedmem",) + extra)
    strips += [StripRenderer(W, H, mesh, env, rank=r, world=world, transport=transport, torch_buffers=True, extra_args=("-sharedmem",) + extra, balance=True, peers=peers) for r in range(world)]
    for s in strips: s.connect_peers(strips)
    for _ in range(StripRenderer.PROFILE_FRAMES): full.frame()
    first_diff = first_report = None
    differing = 0
    for f in range(frames):
        full.frame()
        for s in strips: s.render()
        for s in strips: s.exchange()
        for s in strips:
            for t in strips: s.stream.wait_stream(t.stream); s.stream.wait_stream(t.xstream)
        torch.cuda.synchronize(); full.context.sync()
        a, b = strips[0].context.readback(capi.BUF_BACKBUFFER).reshape(H, W), full.context.readback(capi.BUF_BACKBUFFER).reshape(H, W)
        over = [s.history_overreach(reset=True) for s in strips]
        if any(over) and first_report is None:
            first_report = f
            print("   frame %d: the apron guard reports a history tap %s rows beyond the exchanged rows (per strip)" % (f, over), flush=True)
        if not np.array_equal(a, b):
            differing += 1
            if first_diff is None:
                first_diff = f
                ys, xs = np.nonzero(a != b)
                print("   frame %d: first difference, %d pixels, rows %d..%d, columns %d..%d; bounds %s" % (f, ys.size, ys.min(), ys.max(), xs.min(), xs.max(), strips[0].bounds), flush=True)
    # (the history is recursive: what a reported tap got wrong stays in those pixels for many frames)
    ok = first_diff is None or (not peers and first_report is not None and first_report <= first_diff)
    print("%dx%d %s: %d frames of 8 balanced strips; %s" % (W, H, " ".join(str(x) for x in extra), frames,
          ("all identical to the single context" + ("" if first_report is None else " (history taps beyond the apron from frame %d on: served from the owner's image)" % first_report)) if first_diff is None else
          "identical up to frame %d, then %d frames differ in a few pixels -- %s" % (first_diff - 1, differing, "from the frame on in which the guard reported" if ok else "WITHOUT a report: an error")), flush=True)
    errors += 0 if ok else 1
    comm.destroy(); full.close()
    for s in strips: s.close()
sys.exit(1 if errors else 0)
