"""Multi-process path of the strip renderer on CPU (gloo): the per-frame exchange plan of raytracedggx_amd.strips
delivers every rank's history apron and assembles the frame on rank 0.  The HIP passes themselves are covered by the
GPU tests (two strips against one full frame, tests/test_gpu_parity.py); here the transport is the thing under test."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracedggx_amd.capi import MAX_PEERS
from raytracedggx_amd.strips import HISTORY_APRON, exchange_plan, make_ops, run_exchange, strip_rows

H, W = 120, 16


def _truth(rows):
    """What a correct frame holds in row r, column c (history and back buffer differ by a constant)."""
    r = torch.arange(rows[0], rows[1], dtype=torch.int64)[:, None]
    return r * 1000 + torch.arange(W, dtype=torch.int64)[None, :]


def _worker(rank, world, port, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        b, e = strip_rows(H, rank, world)
        # as in StripRenderer: persistent targets, the operation list built once and reissued every frame
        history = torch.empty((H, W), dtype=torch.int64); back = torch.empty((H, W), dtype=torch.int32)
        tokens = torch.zeros(2 * MAX_PEERS, dtype=torch.int32)
        plan = exchange_plan(H, rank, world)
        ops = make_ops(dist, plan, {"history": history, "backbuffer": back, "token": tokens})
        for frame in range(4):                                   # several frames: the matching must not drift
            history.fill_(-1); back.fill_(-1)                     # rows this rank did not produce are garbage
            tokens[rank] = 100 * rank + frame + 1
            history[b:e] = _truth((b, e)) + frame
            back[b:e] = (_truth((b, e)) + frame + 7).to(torch.int32)
            run_exchange(dist, None, None, ops=ops)
            lo, hi = max(b - HISTORY_APRON, 0), min(e + HISTORY_APRON, H)
            assert torch.equal(history[lo:hi], _truth((lo, hi)) + frame), "rank %d frame %d: history apron" % (rank, frame)
            assert (history[:lo] == -1).all() and (history[hi:] == -1).all(), "nothing beyond the apron is touched"
            if rank == 0:
                assert torch.equal(back, (_truth((0, H)) + frame + 7).to(torch.int32)), "frame assembly on rank 0"
            for op, name, r0, r1, peer in plan:                   # the ordering tokens between ranks that exchange nothing else
                if op == "recv" and name == "token":
                    assert int(tokens[r0]) == 100 * peer + frame + 1, "rank %d frame %d: token of rank %d" % (rank, frame, peer)
        results[rank] = 1
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3, 4])
def test_exchange_over_gloo(world):
    results = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), results), nprocs=world, join=True)
    assert sorted(results.keys()) == list(range(world))


def test_plan_is_consistent():
    """Every send has the matching recv on the peer (same buffer, same rows), in the same per-pair order; strips tile the frame; and
    EVERY ordered pair of ranks has a message (round 4: any rank may read any rank's history image, rtggx_set_history_peers, and the
    exchange is what orders the two)."""
    for height, world in ((1080, 8), (2160, 8), (1080, 3), (120, 2), (1081, 4)):
        plans = [exchange_plan(height, r, world) for r in range(world)]
        rows = [strip_rows(height, r, world) for r in range(world)]
        assert rows[0][0] == 0 and rows[-1][1] == height and all(rows[i][1] == rows[i + 1][0] for i in range(world - 1))
        for a in range(world):
            for b in range(world):
                sends = [(n, r0, r1) if n != "token" else (n, 1) for op, n, r0, r1, peer in plans[a] if op == "send" and peer == b]
                recvs = [(n, r0, r1) if n != "token" else (n, 1) for op, n, r0, r1, peer in plans[b] if op == "recv" and peer == a]
                assert sends == recvs, (height, world, a, b)
                assert a == b or len(sends) >= 1, "no message from rank %d to rank %d" % (a, b)
                assert sum(1 for x in sends if x[0] == "token") == (1 if abs(a - b) >= 2 and b != 0 else 0)
                assert [p for p in exchange_plan(height, a, world, tokens=False)] == [p for p in plans[a] if p[1] != "token"]
        for r in range(world):
            for op, name, r0, r1, peer in plans[r]:
                if name == "token":
                    assert r1 == r0 + 1 and r0 == (r if op == "send" else MAX_PEERS + peer)
                    continue
                assert 0 <= r0 < r1 <= height
                if op == "send":
                    assert rows[r][0] <= r0 and r1 <= rows[r][1], "a rank only sends rows it produced"
                elif name == "history":
                    assert r1 <= rows[r][0] or r0 >= rows[r][1], "and only receives rows it did not"
        single = exchange_plan(height, 0, 1)
        assert single == []


def test_direct_rccl_binding_resolves_every_entry_point():
    """raytracedggx_amd/rccl.py binds the librccl.so torch ships (no GPU needed to load it): every entry point the strip
    exchange calls resolves, and the unique-id structure has the size rccl.h gives it."""
    import ctypes
    from raytracedggx_amd import rccl
    L = rccl.lib()
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclSend", "ncclRecv", "ncclGroupStart", "ncclGroupEnd", "ncclGetErrorString"):
        assert getattr(L, name) is not None
    assert ctypes.sizeof(rccl._UniqueId) == rccl.NCCL_UNIQUE_ID_BYTES == 128
    v = ctypes.c_int()
    L.ncclGetVersion.argtypes = [ctypes.POINTER(ctypes.c_int)]
    assert L.ncclGetVersion(ctypes.byref(v)) == 0 and v.value >= 21800      # ncclSend/ncclRecv to self need >= 2.7
    assert L.ncclGetErrorString(0).decode() != ""


@pytest.mark.parametrize("world", [2, 3, 8])
def test_raw_send_recv_lists_move_the_right_rows(world):
    """strips.plan_to_raw (the argument lists of ncclSend / ncclRecv): carried out with memmove between host arrays, pairing
    the k-th send of rank a to rank b with the k-th receive of rank b from rank a -- RCCL's matching rule inside a group --
    every rank ends up with its neighbours' boundary rows and rank 0 with the whole back buffer."""
    import ctypes
    from raytracedggx_amd import strips
    H, W = 8 * 20 + 3, 6
    hist = [np.zeros((H, W), np.uint64) for _ in range(world)]
    bb = [np.zeros((H, W), np.uint32) for _ in range(world)]
    truth_h = (np.arange(H * W, dtype=np.uint64).reshape(H, W) + 1) * 1000003
    truth_b = ((np.arange(H * W, dtype=np.uint64).reshape(H, W) + 7) * 2654435761 % (2 ** 32)).astype(np.uint32)
    for r in range(world):
        b, e = strips.strip_rows(H, r, world)
        hist[r][b:e] = truth_h[b:e]; bb[r][b:e] = truth_b[b:e]
    tok = [np.zeros(2 * MAX_PEERS, np.uint32) for _ in range(world)]
    for r in range(world):
        tok[r][r] = 1000 + r
    raw = [strips.plan_to_raw(strips.exchange_plan(H, r, world), hist[r].ctypes.data, bb[r].ctypes.data, W, tok[r].ctypes.data) for r in range(world)]
    for a in range(world):
        for b_ in range(world):
            sends = [op for op in raw[a] if op[0] and op[3] == b_]
            recvs = [op for op in raw[b_] if not op[0] and op[3] == a]
            assert len(sends) == len(recvs)
            for (_, sp, sn, _), (_, rp, rn, _) in zip(sends, recvs):
                assert sn == rn
                ctypes.memmove(rp, sp, sn)
    np.testing.assert_array_equal(bb[0], truth_b)
    for r in range(world):
        b, e = strips.strip_rows(H, r, world)
        lo, hi = max(b - strips.HISTORY_APRON, 0), min(e + strips.HISTORY_APRON, H)
        np.testing.assert_array_equal(hist[r][lo:hi], truth_h[lo:hi])
        for p in range(world):      # a token from everybody who sends this rank nothing else
            assert tok[r][MAX_PEERS + p] == (1000 + p if abs(p - r) >= 2 and r != 0 else 0)


def test_balanced_bounds_and_uneven_plans():
    """strips.balanced_bounds: boundaries where the running row cost passes k/N of the total, every strip at least the
    history apron high; exchange plans over such boundaries stay pairwise consistent and move the right rows."""
    import ctypes
    from raytracedggx_amd import strips
    rng = np.random.default_rng(11)
    H, W = 300, 4
    cost = np.concatenate([np.full(120, 0.3), rng.uniform(5.0, 9.0, 100), np.full(80, 1.0)])     # cheap sky, expensive middle
    for world in (2, 3, 8):
        b = strips.balanced_bounds(cost, world)
        assert b[0] == 0 and b[-1] == H and len(b) == world + 1
        assert all(b[k + 1] - b[k] >= strips.HISTORY_APRON for k in range(world))
        sums = [cost[b[k]:b[k + 1]].sum() for k in range(world)]
        if world <= 3:
            assert max(sums) < 1.25 * cost.sum() / world           # balanced where the minimum height does not bind
        assert strips.balanced_bounds(cost, world) == b            # deterministic
        # the plans of all ranks over these boundaries pair up, and carried out they deliver aprons and the whole back buffer
        hist = [np.zeros((H, W), np.uint64) for _ in range(world)]
        bb = [np.zeros((H, W), np.uint32) for _ in range(world)]
        truth_h = (np.arange(H * W, dtype=np.uint64).reshape(H, W) + 1) * 1000003
        truth_b = ((np.arange(H * W, dtype=np.uint64).reshape(H, W) + 7) * 2654435761 % (2 ** 32)).astype(np.uint32)
        for r in range(world):
            hist[r][b[r]:b[r + 1]] = truth_h[b[r]:b[r + 1]]; bb[r][b[r]:b[r + 1]] = truth_b[b[r]:b[r + 1]]
        tok = [np.zeros(2 * MAX_PEERS, np.uint32) for _ in range(world)]
        raw = [strips.plan_to_raw(strips.exchange_plan(H, r, world, bounds=b), hist[r].ctypes.data, bb[r].ctypes.data, W, tok[r].ctypes.data) for r in range(world)]
        for a in range(world):
            for c in range(world):
                sends = [op for op in raw[a] if op[0] and op[3] == c]
                recvs = [op for op in raw[c] if not op[0] and op[3] == a]
                assert [x[2] for x in sends] == [x[2] for x in recvs]
                for (_, sp, sn, _), (_, rp, rn, _) in zip(sends, recvs):
                    ctypes.memmove(rp, sp, sn)
        np.testing.assert_array_equal(bb[0], truth_b)
        for r in range(world):
            lo, hi = max(b[r] - strips.HISTORY_APRON, 0), min(b[r + 1] + strips.HISTORY_APRON, H)
            np.testing.assert_array_equal(hist[r][lo:hi], truth_h[lo:hi])
    with pytest.raises(ValueError):
        strips.balanced_bounds(np.ones(100), 8)                   # 8 strips of 17 rows do not fit in 100
