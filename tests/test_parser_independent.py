"""The product's `.ray` loader against an INDEPENDENT reader (oracle/ray_scene_py.py).

The render-parity tests feed the oracle the scene the product's own loader flattened, so an error in the loader
(transform order, fan order, cube corners, material inheritance, camera basis) is common to both sides there.
Here the same text goes through a Python reader written from the reference's parser sources and shares no code
with the product; primitives are matched by parse order (`prim_id`).  Tolerance: a few float32 ulp of the
coordinates' magnitude (two independent float32 evaluations of the same matrix products) — parity unpinned
against the reference itself, which ships no scene fixtures and cannot be built here.
"""
import numpy as np
import pytest

import pathtracer_amd as P
from oracle.ray_scene_py import RayScene

H = "SBT-raytracer 1.0\n"


def fmt(x):
    return f"{x:.4f}"  # the reference's scalar lexer takes [0-9.\-e] only (Tokenizer.cu:217-225): no '+', no 'E'


def vec(v):
    return "(" + ",".join(fmt(x) for x in v) + ")"


def random_material(rng):
    parts = []
    for word in ("emissive", "diffuse", "reflective", "transmissive", "ambient", "specular"):
        if rng.random() < 0.5:
            parts.append(f"{word} = {vec(rng.random(3))};")
    if rng.random() < 0.4:
        parts.append(f"index = {fmt(1 + rng.random())};")
    if rng.random() < 0.3:
        parts.append(f"shininess = {fmt(rng.random() * 50)};")
    return "{ " + " ".join(parts) + " }"


def random_primitive(rng):
    kind = rng.choice(["sphere", "box", "square", "polymesh", "trimesh"])
    mat = f"material = {random_material(rng)};" if rng.random() < 0.6 else ""
    if kind in ("sphere", "box", "square"):
        return f"{kind} {{ {mat} }}"
    n = int(rng.integers(3, 8))
    pts = rng.uniform(-1, 1, (n, 3))
    faces = []
    for _ in range(int(rng.integers(1, 4))):
        k = int(rng.integers(3, min(n, 6) + 1))
        faces.append("(" + ",".join(str(int(i)) for i in rng.permutation(n)[:k]) + ")")
    body = f"points = ({','.join(vec(p) for p in pts)}); faces = ({','.join(faces)});"
    if rng.random() < 0.4:  # per-vertex materials inherit from the mesh material (Parser.cu:483)
        body = mat + f" materials = ({','.join(random_material(rng) for _ in range(n))}); " + body
    else:
        body = body + " " + mat
    return f"{kind} {{ {body} }}"


def random_element(rng, depth):
    r = rng.random()
    if depth == 0 or r < 0.3:
        return random_primitive(rng)
    if r < 0.42:
        return "{ " + " ".join(random_element(rng, depth - 1) for _ in range(int(rng.integers(1, 4)))) + " }"
    child = random_element(rng, depth - 1)
    if r < 0.58:
        return f"translate({','.join(fmt(x) for x in rng.uniform(-3, 3, 3))}, {child})"
    if r < 0.72:
        return f"scale({fmt(rng.uniform(0.3, 2))}, {child})"
    if r < 0.82:
        return f"scale({','.join(fmt(x) for x in rng.uniform(0.3, 2, 3))}, {child})"
    if r < 0.94:
        axis = rng.uniform(-1, 1, 3)
        return f"rotate({','.join(fmt(x) for x in axis)},{fmt(rng.uniform(-3, 3))}, {child})"
    m = np.eye(4)
    m[:3, :] = rng.uniform(-1.5, 1.5, (3, 4))
    return "transform(" + ",".join(vec(row) for row in m) + ", " + child + ")"


def random_scene(seed):
    rng = np.random.default_rng(seed)
    view = rng.uniform(-1, 1, 3)
    up = rng.uniform(-1, 1, 3)
    txt = H + f"camera {{ position = {vec(rng.uniform(-5, 5, 3))}; viewdir = {vec(view)}; updir = {vec(up)}; fov = {fmt(rng.uniform(20, 90))}; aspectratio = {fmt(rng.uniform(0.5, 2))}; }}\n"
    for _ in range(int(rng.integers(2, 7))):
        r = rng.random()
        if r < 0.15:
            txt += f"material = {random_material(rng)};\n"  # inherited by everything that follows (Parser.cu:59-61)
        elif r < 0.25:
            txt += f"point_light {{ position = {vec(rng.uniform(-4, 4, 3))}; colour = {vec(rng.random(3))}; constant_attenuation_coeff = 0.25; }}\n"
        elif r < 0.3:
            txt += "ambient_light { color = (0.1000,0.1000,0.1000); }\ndirectional_light { direction = (0,-1,0); colour = (1,1,1); }\n"
        else:
            txt += random_element(rng, 3) + "\n"
    return txt


def product_prims(sc):
    """The product's primitives in parse order: ("tri", verts[3,3], [rows]) / ("sph", centre, radius, row)."""
    s = sc.soa.contents
    n = s.n_tri + s.n_sph
    prim_id = sc.array("prim_id", (n,), np.int32)
    mats = sc.array("materials", (s.n_mat, 20), np.float32)
    row = lambda m: np.concatenate([mats[m, 0:3], mats[m, 4:7], mats[m, 8:11], mats[m, 12:15], mats[m, 16:17]])
    verts = sc.slot_vertices()[:, :, :3]
    tri_mat = sc.array("tri_mat", (s.n_tri, 3), np.uint32)
    has_v = sc.array("tri_has_vmat", (s.n_tri,), np.uint8)
    sph = sc.array("sph_center_radius", (s.n_sph, 4), np.float32)
    sph_mat = sc.array("sph_mat", (s.n_sph,), np.uint32)
    out = [None] * n
    assert sorted(prim_id.tolist()) == list(range(n)), "prim_id is a permutation of the parse order"
    for slot in range(n):
        if slot < s.n_tri:
            rows = [row(m) for m in tri_mat[slot]] if has_v[slot] else [row(tri_mat[slot][0])]
            out[prim_id[slot]] = ("tri", verts[slot], rows)
        else:
            k = slot - s.n_tri
            out[prim_id[slot]] = ("sph", sph[k, :3], sph[k, 3], row(sph_mat[k]))
    return out


def close(a, b, scale):
    return np.all(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)) <= 2e-6 * scale)


@pytest.mark.parametrize("seed", range(60))
def test_loader_agrees_with_the_independent_reader(seed):
    txt = random_scene(seed)
    ref = RayScene(txt)
    sc = P.Scene.from_text(txt)
    got = product_prims(sc)
    # a face with two equal corners is dropped by TrimeshFace::create (trimesh.cu:17-21); random corners never coincide
    assert len(got) == len(ref.prims), txt
    for k, (g, r) in enumerate(zip(got, ref.prims)):
        assert g[0] == r[0], (k, txt)
        if g[0] == "tri":
            scale = max(1.0, float(np.abs(r[1]).max()))
            assert close(g[1], r[1], scale), (k, g[1], r[1], txt)
            assert len(g[2]) == len(r[2]), (k, txt)
            for gm, rm in zip(g[2], r[2]):
                assert np.array_equal(gm, rm.row()), (k, gm, rm.row(), txt)
        else:
            scale = max(1.0, float(np.abs(r[1]).max()))
            assert close(g[1], r[1], scale) and close(g[2], r[2], max(1.0, abs(float(r[2])))), (k, g, r, txt)
            assert np.array_equal(g[3], r[3].row()), (k, txt)
    s = sc.soa.contents
    eye, u, v, look = ref.camera_basis()
    for name, want in (("cam_eye", eye), ("cam_u", u), ("cam_v", v), ("cam_look", look)):
        have = np.array(getattr(s, name)[:3], np.float32)
        assert close(have, want, max(1.0, float(np.abs(want).max()))), (name, have, want)


def test_the_scenes_the_benchmarks_use():
    from pathtracer_amd import scenes
    for txt in (scenes.simple_ray(), scenes.cornell_ray(), scenes.glass_ray()):
        ref, sc = RayScene(txt), P.Scene.from_text(txt)
        got = product_prims(sc)
        assert len(got) == len(ref.prims)
        for g, r in zip(got, ref.prims):
            assert g[0] == r[0]
            assert close(g[1], r[1], max(1.0, float(np.abs(r[1]).max())))
            if g[0] == "sph":
                assert close(g[2], r[2], 1.0) and np.array_equal(g[3], r[3].row())
            else:
                assert all(np.array_equal(gm, rm.row()) for gm, rm in zip(g[2], r[2]))
