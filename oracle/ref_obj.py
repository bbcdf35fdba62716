"""ORACLE / TEST INFRASTRUCTURE ONLY.  ctypes view of oracle/_ref/libobjloader_ref.so: the REFERENCE's own OBJ importer
(XUSG::ObjLoader, /root/reference/RayTracedGGX/XUSG/Optional/XUSGObjLoader.cpp) built by `make -C oracle ref` in the build
container (oracle/ref_objloader.cpp says how).  The library travels to the GPU box with the snapshot; the reference does not.
Only tests/ and tests/golden/make_obj_golden.py import this."""
import ctypes as C
import os

import numpy as np

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libobjloader_ref.so")
_lib = None


def available():
    return os.path.exists(LIB_PATH)


def _load():
    global _lib
    if _lib is None:
        L = C.CDLL(LIB_PATH)
        L.ref_obj_import.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_void_p]
        L.ref_obj_copy.argtypes = [C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def obj_import(path):
    """XUSG::ObjLoader::Import(path, needNorm=true, needAABB=true) -> (verts[nv, 6] float32, indices[ni] uint32, aabb[6])."""
    L = _load()
    nv, ni, stride = C.c_uint32(), C.c_uint32(), C.c_uint32()
    aabb = np.zeros(6, np.float32)
    if L.ref_obj_import(os.fsencode(path), C.byref(nv), C.byref(ni), C.byref(stride), aabb.ctypes.data_as(C.c_void_p)) != 0:
        raise IOError("reference ObjLoader::Import failed for %s" % path)
    # 24 bytes (position, normal); 32 when the file has `vt` records: the reference then appends an (unfilled) float2 per
    # vertex (XUSGObjLoader.cpp:160) -- a layout the sample never uses (its meshes have none and its shaders read 24-byte
    # vertices).  Position and normal are the first 24 bytes either way; that is what is returned.
    assert stride.value in (24, 32)
    raw = np.zeros((nv.value, stride.value // 4), np.float32)
    i = np.zeros(ni.value, np.uint32)
    L.ref_obj_copy(raw.ctypes.data_as(C.c_void_p), i.ctypes.data_as(C.c_void_p))
    return np.ascontiguousarray(raw[:, :6]), i, aabb
