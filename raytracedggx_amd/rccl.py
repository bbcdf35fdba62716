"""RCCL point-to-point, called directly (ctypes over the librccl.so that torch.distributed already has in the process).

The strip exchange is 3-9 small sends/receives per frame, every frame (strips.exchange_plan).  Issued through
torch.distributed's P2POp batch each of them costs ~10 us of host time on top of ~25 us for the batch (measured on
MI355X: 6 ops 74 us, 16 ops 160 us), which at 8 strips is more than the GPU needs for the whole strip (~95 us).  The same
ncclSend/ncclRecv calls made from here, inside one ncclGroupStart/End on the renderer's own stream, cost a few
microseconds each.  torch.distributed still does the rendezvous (it carries the ncclUniqueId to the other ranks) and
everything outside the frame loop (barriers, the reductions of bench.py).
"""
import ctypes
import os

NCCL_UNIQUE_ID_BYTES = 128          # rccl.h:40
ncclUint8 = 1                       # rccl.h:460


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * NCCL_UNIQUE_ID_BYTES)]


_lib = None


def lib():
    """librccl.so: the copy torch loaded (two copies of RCCL in one process would each bring their own state)."""
    global _lib
    if _lib is None:
        import torch
        candidates = [os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "/opt/rocm/lib/librccl.so"]
        path = next((p for p in candidates if os.path.exists(p)), None)
        if path is None:
            raise ImportError("librccl.so not found (looked in %s)" % ", ".join(candidates))
        L = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        L.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
        L.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _UniqueId, ctypes.c_int]
        L.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        L.ncclSend.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.ncclRecv.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.ncclGroupStart.argtypes = []
        L.ncclGroupEnd.argtypes = []
        L.ncclGetErrorString.argtypes = [ctypes.c_int]
        L.ncclGetErrorString.restype = ctypes.c_char_p
        for f in (L.ncclGetUniqueId, L.ncclCommInitRank, L.ncclCommDestroy, L.ncclSend, L.ncclRecv, L.ncclGroupStart, L.ncclGroupEnd):
            f.restype = ctypes.c_int
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %s" % (what, lib().ncclGetErrorString(rc).decode()))


class Communicator:
    """One RCCL communicator over the ranks of `dist`'s default group; the current HIP device must already be set."""

    def __init__(self, dist, rank, world):
        import torch
        self.rank, self.world = rank, world
        self.handle = ctypes.c_void_p()
        uid = _UniqueId()
        ok, why = 1, ""
        try:
            L = lib()
            if rank == 0:
                _check(L.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
        except Exception as e:             # the ranks must still meet in the broadcast below, and all must learn of it
            ok, why = 0, repr(e)
        if world > 1:
            t = torch.frombuffer(bytearray(bytes([ok]) + ctypes.string_at(ctypes.byref(uid), NCCL_UNIQUE_ID_BYTES)), dtype=torch.uint8).cuda()
            dist.broadcast(t, src=0)
            raw = t.cpu().numpy().tobytes()
            flag = torch.tensor([ok and raw[0]], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                raise RuntimeError("direct RCCL exchange unavailable on at least one rank (%s)" % (why or "see the other ranks"))
            ctypes.memmove(ctypes.byref(uid), raw[1:], NCCL_UNIQUE_ID_BYTES)
        elif not ok:
            raise RuntimeError(why)
        _check(L.ncclCommInitRank(ctypes.byref(self.handle), world, uid, rank), "ncclCommInitRank")

    def exchange(self, ops, stream):
        """ops: [(is_send, device_pointer, nbytes, peer)] -- one group launch on `stream` (a hipStream_t as an integer)."""
        if not ops:
            return
        L = lib()
        s = ctypes.c_void_p(stream)
        _check(L.ncclGroupStart(), "ncclGroupStart")
        for is_send, ptr, nbytes, peer in ops:
            rc = (L.ncclSend if is_send else L.ncclRecv)(ptr, nbytes, ncclUint8, peer, self.handle, s)
            if rc != 0:
                L.ncclGroupEnd()
                _check(rc, "ncclSend" if is_send else "ncclRecv")
        _check(L.ncclGroupEnd(), "ncclGroupEnd")

    def destroy(self):
        if self.handle:
            lib().ncclCommDestroy(self.handle)
            self.handle = ctypes.c_void_p()
