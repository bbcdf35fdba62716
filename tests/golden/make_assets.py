"""Generates tests/golden/assets.npz from the data assets of the reference (run in the build
container only; /root/reference does not exist on the GPU box).

What is stored is data, not source: for each mesh the raw `v` records (float32) and the 1-based `f`
triangles of the OBJ file, and the bytes of the DDS cube map -- the inputs BASELINE.json's configs
name (bunny.obj, dragon.obj, rnl_cross.dds).  tests/assets.py writes equivalent .obj/.dds files into
a cache directory so that the product's own loaders (and the oracle's) read them from disk.
Also stores the known-answer facts of SURVEY.md 8c (counts, AABB, first indices, FNV-1a hashes) that
pin both OBJ importers.
"""
import json
import os
import sys

import numpy as np

REF = "/root/reference/Bin/Assets"
HERE = os.path.dirname(os.path.abspath(__file__))


def parse_obj(path):
    v, f = [], []
    for line in open(path):
        t = line.split()
        if not t:
            continue
        if t[0] == "v":
            v.append([np.float32(x) for x in t[1:4]])
        elif t[0] == "f":
            idx = [int(x.split("/")[0]) for x in t[1:]]
            assert len(idx) == 3, "fixture generator expects triangles"
            f.append(idx)
    return np.asarray(v, np.float32), np.asarray(f, np.int32)


def main():
    out = {}
    for name in ("bunny", "dragon"):
        v, f = parse_obj(os.path.join(REF, name + ".obj"))
        out[name + "_v"], out[name + "_f"] = v, f
        print(name, v.shape, f.shape)
    out["rnl_cross_dds"] = np.fromfile(os.path.join(REF, "rnl_cross.dds"), np.uint8)
    np.savez_compressed(os.path.join(HERE, "assets.npz"), **out)
    # golden facts of ObjLoader::Import (little-endian FNV-1a-32 over the raw arrays)
    facts = {
        "bunny": {"num_verts": 34835, "num_indices": 208998, "aabb_min": [-5.0151, -0.0442, -3.8870], "aabb_max": [5.0151, 9.8982, 3.8870],
                  "first_indices": [34834, 33422, 12706, 34834, 12706, 22064], "v0": [1.4870, 0.3736, -2.2576],
                  "n0": [-0.238448, -0.895268, 0.376347], "fnv_verts": "6e718c4b", "fnv_indices": "96da0f35"},
        "dragon": {"num_verts": 50000, "num_indices": 300000, "aabb_min": [-7.0467, 0.0, -3.1513], "aabb_max": [7.0467, 9.9399, 3.1513],
                   "first_indices": [47437, 42256, 29824, 29823, 47437, 29824], "fnv_verts": "1caf093d", "fnv_indices": "f0ba71fa"},
        "TuringBowl": {"num_verts": 23188, "num_indices": 68232, "fnv_verts": "72feea4e", "fnv_indices": "33ee44fe"},
    }
    json.dump(facts, open(os.path.join(HERE, "obj_import.json"), "w"), indent=1)


if __name__ == "__main__":
    sys.exit(main())
