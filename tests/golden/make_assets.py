"""Generates tests/golden/assets.npz from the data assets of the reference (run in the build
container only; /root/reference does not exist on the GPU box).

What is stored is data, not source: for each mesh the raw `v` (and `vn`) records (float32) and the 1-based `f`
triangles of the OBJ file, and the bytes of the DDS cube map -- the inputs BASELINE.json's configs and the
reference's Bin/*.bat files name (bunny.obj, dragon.obj, TuringBowl.obj, rnl_cross.dds).  tests/assets.py writes
equivalent .obj/.dds files into a cache directory so that the product's own loaders (and the oracle's) read them
from disk.  The known-answer facts of ObjLoader::Import come from tests/golden/make_obj_golden.py, which also
checks that the regenerated files import exactly like the originals.
"""
import os
import sys

import numpy as np

REF = "/root/reference/Bin/Assets"
HERE = os.path.dirname(os.path.abspath(__file__))


def parse_obj(path):
    """v / vn records and triangles.  Faces come back as int32 [T, 3, 2]: (position index, normal index), 1-based,
    normal index 0 where the file gives none ("f a b c")."""
    v, vn, f = [], [], []
    for line in open(path):
        t = line.split()
        if not t:
            continue
        if t[0] == "v":
            v.append([np.float32(x) for x in t[1:4]])
        elif t[0] == "vn":
            vn.append([np.float32(x) for x in t[1:4]])
        elif t[0] == "vt":
            raise ValueError("fixture generator expects no texture coordinates")
        elif t[0] == "f":
            assert len(t) == 4, "fixture generator expects triangles"
            tri = []
            for corner in t[1:]:
                p = corner.split("/")
                tri.append([int(p[0]), int(p[2]) if len(p) == 3 and p[2] else 0])
            f.append(tri)
    return np.asarray(v, np.float32), np.asarray(vn, np.float32).reshape(-1, 3), np.asarray(f, np.int32)


def main():
    out = {}
    for name in ("bunny", "dragon", "TuringBowl"):
        v, vn, f = parse_obj(os.path.join(REF, name + ".obj"))
        out[name + "_v"] = v
        if len(vn):
            out[name + "_vn"], out[name + "_f"] = vn, f            # [T, 3, 2]
        else:
            out[name + "_f"] = np.ascontiguousarray(f[:, :, 0])     # [T, 3]
        print(name, v.shape, vn.shape, f.shape)
    out["rnl_cross_dds"] = np.fromfile(os.path.join(REF, "rnl_cross.dds"), np.uint8)
    np.savez_compressed(os.path.join(HERE, "assets.npz"), **out)


if __name__ == "__main__":
    sys.exit(main())
