"""Host side: .ray grammar (incl. the reference's quirks), scene flattening, BVH."""
import numpy as np
import pytest

import pathtracer_amd as P
from pathtracer_amd import scenes

H = scenes.HEADER


def soa_arrays(sc):
    s = sc.soa.contents
    n = s.n_tri + s.n_sph
    return dict(
        nodes=sc.array("nodes", (s.n_nodes, 16), np.float32), prim_id=sc.array("prim_id", (n,), np.int32),
        sph=sc.array("sph_center_radius", (s.n_sph, 4), np.float32), mats=sc.array("materials", (s.n_mat, 20), np.float32),
        tri_mat=sc.array("tri_mat", (s.n_tri, 3), np.uint32), sph_mat=sc.array("sph_mat", (s.n_sph,), np.uint32),
        has_vmat=sc.array("tri_has_vmat", (s.n_tri,), np.uint8), v0=sc.array("tri_v0", (s.n_tri, 4), np.float32))


def test_sphere_position_and_radius_come_from_the_transform():
    # Parser.cu:338-340: centre = xform*(0,0,0,1), radius = (xform*(1,0,0,0)).x
    sc = P.Scene.from_text(H + "translate(1,2,3, scale(2,5,7, sphere { }))\n")
    a = soa_arrays(sc)
    assert a["sph"].tolist() == [[1, 2, 3, 2]]
    assert a["mats"][0].tolist() == [0] * 16 + [1, 1, 1, 1]   # default material, material.h:129-133


def test_point_light_is_an_emissive_transmissive_unit_sphere():
    sc = P.Scene.from_text(H + "point_light { position = (1,2,3); colour = (0.5,0.25,1); constant_attenuation_coeff = 0.2; }\n")
    a = soa_arrays(sc)
    assert a["sph"].tolist() == [[1, 2, 3, 1]]
    assert a["mats"][0].tolist() == [0.5, 0.25, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 0, 1, 1, 1, 0]  # Parser.cu:650


def test_box_and_square_are_twelve_triangles():
    sc = P.Scene.from_text(H + "box { }\nsquare { }\n")
    s = sc.soa.contents
    assert s.n_tri == 24 and s.n_sph == 0
    v = sc.slot_vertices()[:, :, :3]
    ids = soa_arrays(sc)["prim_id"]
    box = v[np.argsort(ids)][:12].reshape(-1, 3)
    sq = v[np.argsort(ids)][12:].reshape(-1, 3)
    assert np.abs(box).max() == 0.5 and set(np.unique(np.abs(box))) == {0.5}
    assert set(np.unique(np.abs(sq[:, 2]))) == {np.float32(0.001)}     # trimesh.cu:162-173


def test_fan_triangulation_and_per_vertex_materials():
    txt = H + """polymesh { points = ((0,0,0),(1,0,0),(1,1,0),(0,1,0),(0.5,2,0));
      faces = ((0,1,2,3,4));
      materials = ({ diffuse = (1,0,0); }, { diffuse = (0,1,0); }, { diffuse = (0,0,1); }, { diffuse = (1,1,0); }, { diffuse = (0,1,1); }); }\n"""
    sc = P.Scene.from_text(txt)
    a = soa_arrays(sc)
    assert sc.soa.contents.n_tri == 3 and a["has_vmat"].tolist() == [1, 1, 1]   # Parser.cu:580-596
    order = np.argsort(a["prim_id"])
    kd = a["mats"][:, 4:7]
    got = [[kd[m].tolist() for m in a["tri_mat"][k]] for k in order]
    assert got[0] == [[1, 0, 0], [0, 1, 0], [0, 0, 1]] and got[1] == [[1, 0, 0], [0, 0, 1], [1, 1, 0]]


def test_reference_quirks_kept():
    # 'polymesh' == 'trimesh', 'colour' == 'color', comments, map(...) -> (1,0,1), cylinder/cone ignored
    txt = H + """/* block */ // line
    cylinder { } cone { height = 2; capped = true; }
    trimesh { material = { diffuse = map("foo.png"); index = 1.3; }; points = ((0,0,0),(1,0,0),(0,1,0)); faces = ((0,1,2)); }
    directional_light { direction = (0,-1,0); color = (1,1,1); } ambient_light { colour = (0.1,0.1,0.1); }\n"""
    sc = P.Scene.from_text(txt)
    a = soa_arrays(sc)
    assert sc.n_prims == 1
    assert a["mats"][0][4:8].tolist() == [1, 0, 1, 0] and a["mats"][0][16:].tolist() == [np.float32(1.3)] * 4
    # quaternian's 4th component is dropped (Parser.cu:788-799)
    q1 = P.Scene.from_text(H + "camera { quaternian = (1,0,0,0); }\n").soa.contents
    q2 = P.Scene.from_text(H + "camera { quaternian = (1,0,0,0.7); }\n").soa.contents
    assert list(q1.cam_look) == list(q2.cam_look) and list(q1.cam_u) == list(q2.cam_u)
    # `transform` rows lose their 4th entry the same way -> no translation
    t = P.Scene.from_text(H + "transform((1,0,0,5),(0,1,0,6),(0,0,1,7),(0,0,0,1), sphere { })\n")
    assert soa_arrays(t)["sph"][0][:3].tolist() == [0, 0, 0]


@pytest.mark.parametrize("txt,msg", [
    ("SBT-raytracer 2.0\n", "too high"),
    (H + "sphere { bogus = 1; }", "Expected: sphere attributes"),
    (H + "{ material = { diffuse = (1,1,1); } sphere { } }", "Expected: '}' or geometry"),   # Parser.cu:178-183 fall-through
    (H + "camera { viewdir = (0,0,-1); }", "Expected: 'updir'"),
    (H + "polymesh { points = ((0,0,0),(1,0,0),(0,1,0)); faces = ((0,1,3)); }", "Bad face in trimesh"),
    (H + "polymesh { points = ((0,0,0),(1,0,0),(0,1,0)); faces = ((0,1)); }", "at least 3 vertices"),
    (H + "/* never closed", "Unterminated comment"),
    (H + "point_light { position = (0,0,0); }", "Expected: 'color'"),
    (H + "sphere { } @", "unexpected character"),
])
def test_parse_errors(txt, msg):
    with pytest.raises(ValueError, match=msg.replace("(", r"\(").replace(")", r"\)")):
        P.Scene.from_text(txt)


def test_text_and_programmatic_scene_are_identical():
    a = P.Scene.from_text(scenes.blob_ray(2))
    b = scenes.blob_scene(2)
    sa, sb = a.soa.contents, b.soa.contents
    assert (sa.n_tri, sa.n_nodes, sa.root_ref) == (sb.n_tri, sb.n_nodes, sb.root_ref)
    for name, shape in (("nodes", (sa.n_nodes, 16)), ("tri_v0", (sa.n_tri, 4)), ("tri_ninv", (sa.n_tri, 4)), ("materials", (sa.n_mat, 20))):
        assert np.array_equal(a.array(name, shape, np.float32).view(np.uint32), b.array(name, shape, np.float32).view(np.uint32)), name
    assert list(sa.cam_u) == list(sb.cam_u) and list(sa.cam_look) == list(sb.cam_look)


@pytest.mark.parametrize("name", ["simple", "cornell", "glass", "blob3"])
def test_bvh_structure(scenes_small, name):
    sc = scenes_small[name]
    s = sc.soa.contents
    a = soa_arrays(sc)
    n = s.n_tri + s.n_sph
    assert s.n_nodes == n - 1 and sorted(a["prim_id"].tolist()) == list(range(n))
    refs = a["nodes"][:, 12:14].copy().view(np.int32)
    verts = sc.slot_vertices()[:, :, :3]
    seen = np.zeros(n, int)

    def bounds(ref):  # returns (lo, hi, depth) of the subtree, checking containment on the way
        if ref < 0:
            slot = ~ref
            seen[slot] += 1
            if slot < s.n_tri:
                return verts[slot].min(0), verts[slot].max(0), 0
            c, r = a["sph"][slot - s.n_tri][:3], abs(a["sph"][slot - s.n_tri][3])
            return c - r, c + r, 0
        nd = a["nodes"][ref]
        out = []
        for c in range(2):
            lo, hi, d = bounds(int(refs[ref, c]))
            blo, bhi = nd[6 * c:6 * c + 3], nd[6 * c + 3:6 * c + 6]
            assert np.all(blo <= lo) and np.all(bhi >= hi), "child box does not contain its subtree"
            out.append((blo, bhi, d))
        return np.minimum(out[0][0], out[1][0]), np.maximum(out[0][1], out[1][1]), 1 + max(out[0][2], out[1][2])

    _, _, depth = bounds(int(s.root_ref))
    assert np.all(seen == 1) and depth == sc.bvh_depth <= 64


def test_single_primitive_and_empty_scene():
    one = P.Scene.from_text(H + "sphere { }\n").soa.contents
    assert (one.n_nodes, one.root_ref) == (0, -1)      # leaf ref ~0 (bvh.cu:182-183)
    assert P.Scene.from_text(H).n_prims == 0


def test_large_mesh_builds_with_bounded_depth():
    sc = scenes.blob_scene(5)
    assert sc.n_prims == 20 * 4 ** 5 + 2 and sc.bvh_depth <= 40 and sc.sah_cost > 0
    # many coincident primitives: the median-split guard keeps the tree shallow
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    f = np.tile(np.array([[0, 1, 2]], np.int32), (5000, 1))
    sc = P.Scene.from_mesh(v, f, scenes.mat20(diffuse=(1, 1, 1)), scenes.BLOB_CAMERA)
    assert sc.n_prims == 5000 and sc.bvh_depth <= 16
