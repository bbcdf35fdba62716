"""Writes tests/golden/obj_import.json: known answers of the REFERENCE's OBJ importer (XUSG::ObjLoader::Import,
/root/reference/RayTracedGGX/XUSG/Optional/XUSGObjLoader.cpp) for the three shipped meshes.

Build container only.  It (1) builds oracle/_ref/libobjloader_ref.so from the reference source where it lies
(`make -C oracle ref`: oracle/Makefile, oracle/ref_objloader.cpp), (2) imports Bin/Assets/{bunny,dragon,TuringBowl}.obj
with it, (3) imports the .obj files tests/assets.py regenerates from tests/golden/assets.npz and insists that they give
byte-identical vertex and index arrays (so tests on the GPU box, where /root/reference does not exist, read equivalent
files), and (4) records counts, AABB, first indices / vertex and the little-endian FNV-1a-32 of the raw arrays.
These are the values SURVEY.md 8c quotes; nothing here is typed in by hand.
"""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF_ASSETS = "/root/reference/Bin/Assets"


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL)
    import assets
    from oracle import ref_obj
    facts = {}
    for name in ("bunny", "dragon", "TuringBowl"):
        v, i, aabb = ref_obj.obj_import(os.path.join(REF_ASSETS, name + ".obj"))
        v2, i2, aabb2 = ref_obj.obj_import(assets.path(name + ".obj"))
        assert np.array_equal(v.view(np.uint32), v2.view(np.uint32)) and np.array_equal(i, i2) and np.array_equal(aabb, aabb2), \
            "%s: the regenerated .obj does not import like the original" % name
        facts[name] = {
            "num_verts": int(v.shape[0]), "num_indices": int(i.size),
            "aabb_min": [round(float(x), 4) for x in aabb[:3]], "aabb_max": [round(float(x), 4) for x in aabb[3:]],
            "first_indices": [int(x) for x in i[:6]],
            "v0": [round(float(x), 4) for x in v[0, :3]], "n0": [round(float(x), 6) for x in v[0, 3:]],
            "fnv_verts": "%08x" % assets.fnv1a32(v.tobytes()), "fnv_indices": "%08x" % assets.fnv1a32(i.tobytes()),
        }
        print(name, facts[name])
    with open(os.path.join(HERE, "obj_import.json"), "w") as f:
        json.dump(facts, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    sys.exit(main())
