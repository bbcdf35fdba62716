"""The product's OBJ importer (host/ObjLoader.cpp) and the oracle's (oracle/orc_obj.h) against the REFERENCE's own importer,
XUSG::ObjLoader, compiled from /root/reference/RayTracedGGX/XUSG/Optional/XUSGObjLoader.cpp into
oracle/_ref/libobjloader_ref.so (oracle/Makefile target `ref`, build container only; the built library travels to the GPU
box).  This is the one row of the hot path where the reference's code itself can be run here: vertex and
index arrays are compared bit for bit -- on the three meshes the reference ships (Bin/Assets, regenerated from
tests/golden/assets.npz) and on generated files covering every face syntax the importer parses."""
import json
import os

import numpy as np
import pytest

import assets
from oracle import oracle as O
from oracle import ref_obj

HERE = os.path.dirname(os.path.abspath(__file__))

needs_ref = pytest.mark.skipif(not ref_obj.available(), reason="oracle/_ref/libobjloader_ref.so not built (needs /root/reference: make -C oracle ref)")


def _same(a, b, what):
    np.testing.assert_array_equal(a[0].view(np.uint32), b[0].view(np.uint32), err_msg=what + ": vertices")
    np.testing.assert_array_equal(a[1], b[1], err_msg=what + ": indices")
    np.testing.assert_array_equal(a[2].view(np.uint32), b[2].view(np.uint32), err_msg=what + ": AABB")


@needs_ref
@pytest.mark.parametrize("name", ["bunny", "dragon", "TuringBowl"])
def test_shipped_meshes_import_like_the_reference(built, name):
    """bunny / dragon: no normals in the file (face-normal synthesis, XUSGObjLoader.cpp:337-384); TuringBowl: v//vn faces with
    per-corner normals and vertex splitting (:300-335; Bin/TuringBowl.bat)."""
    from raytracedggx_amd import app
    path = assets.path(name + ".obj")
    ref = ref_obj.obj_import(path)
    _same(app.obj_import(path), ref, "product " + name)
    _same(O.obj_import(path), ref, "oracle " + name)
    # and the committed known answers are what the reference's importer says today (tests/golden/make_obj_golden.py)
    f = json.load(open(os.path.join(HERE, "golden", "obj_import.json")))[name]
    assert ref[0].shape == (f["num_verts"], 6) and ref[1].size == f["num_indices"]
    assert "%08x" % assets.fnv1a32(ref[0].tobytes()) == f["fnv_verts"] and "%08x" % assets.fnv1a32(ref[1].tobytes()) == f["fnv_indices"]


def _generated_obj(rng, syntax, polygons, negative):
    """A random mesh as OBJ text: `syntax` in {"v", "v/vt", "v//vn", "v/vt/vn"}; faces are triangles, or polygons of 3-6
    corners (fan triangulation, :267-297); indices 1-based or negative (relative, :243)."""
    nv, nn, nt = int(rng.integers(6, 40)), int(rng.integers(1, 12)), int(rng.integers(1, 9))
    lines = ["# generated", "o thing"]
    lines += ["v %.6f %.6f %.6f" % tuple(rng.uniform(-5, 5, 3)) for _ in range(nv)]
    if "vt" in syntax:
        lines += ["vt %.4f %.4f" % tuple(rng.uniform(0, 1, 2)) for _ in range(nt)]
    if "vn" in syntax:
        for _ in range(nn):
            n = rng.standard_normal(3)
            lines.append("vn %.6f %.6f %.6f" % tuple(n / np.linalg.norm(n) * rng.uniform(0.5, 2.0)))
    lines.append("s 1")
    for _ in range(int(rng.integers(1, 30))):
        k = int(rng.integers(3, 7)) if polygons else 3
        corners = []
        for c in rng.choice(nv, k, replace=False):
            v = int(c) - nv if negative else int(c) + 1
            t = int(rng.integers(0, nt)); t = t - nt if negative else t + 1
            n = int(rng.integers(0, nn)); n = n - nn if negative else n + 1
            corners.append({"v": "%d" % v, "v/vt": "%d/%d" % (v, t), "v//vn": "%d//%d" % (v, n), "v/vt/vn": "%d/%d/%d" % (v, t, n)}[syntax])
        lines.append("f " + " ".join(corners))
    return "\n".join(lines) + "\n"


@needs_ref
@pytest.mark.parametrize("syntax", ["v", "v/vt", "v//vn", "v/vt/vn"])
@pytest.mark.parametrize("polygons", [False, True], ids=["triangles", "polygons"])
@pytest.mark.parametrize("negative", [False, True], ids=["absolute", "relative"])
def test_generated_files_import_like_the_reference(built, tmp_path, syntax, polygons, negative):
    from raytracedggx_amd import app
    rng = np.random.default_rng(["v", "v/vt", "v//vn", "v/vt/vn"].index(syntax) * 4 + polygons * 2 + negative)
    for k in range(12):
        p = tmp_path / ("g%d.obj" % k)
        p.write_text(_generated_obj(rng, syntax, polygons, negative))
        ref = ref_obj.obj_import(str(p))
        assert ref[1].size >= 3
        _same(app.obj_import(str(p)), ref, "product %s #%d" % (syntax, k))
        _same(O.obj_import(str(p)), ref, "oracle %s #%d" % (syntax, k))
