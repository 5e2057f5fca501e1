"""Scenes that stress the 16-bit plane codes of the quantised 4-wide nodes (DevScene::qnodes):
the code grid spans the scene box, so small objects in a huge scene, scenes far from the
coordinate origin, flat scenes (zero extent on an axis) and cameras far outside the scene
are where an outward-rounding or margin mistake would cull a hit.  Every case is compared
with the oracle bit for bit (render) and with its test of every primitive (ray probes);
host-built and device-built trees."""
import numpy as np
import pytest

import pathtracer_amd as P
from pathtracer_amd import scenes
from oracle import oracle_py as O

pytestmark = pytest.mark.gpu
H = scenes.HEADER


def same(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def cam(pos, target=(0, 0.3, 0), fov=40):
    return "camera { position = (%.9g,%.9g,%.9g); viewdir = (%.9g,%.9g,%.9g); updir = (0,1,0); fov = %g; aspectratio = 1.3; }\n" % (
        *pos, target[0] - pos[0], target[1] - pos[1], target[2] - pos[2], fov)


def spheres(rng, n, centre, spread, radius):
    body = ""
    for _ in range(n):
        c = centre + rng.uniform(-spread, spread, 3)
        kind = int(rng.integers(0, 3))
        mat = ["diffuse = (0.7,0.6,0.5);", "diffuse = (0.8,0.8,0.8); reflective = (0.9,0.9,0.9);",
               "diffuse = (0.1,0.1,0.1); transmissive = (0.9,0.9,0.9); index = 1.4;"][kind]
        body += "translate(%.9g,%.9g,%.9g, scale(%.9g, sphere { material = { %s } }))\n" % (c[0], c[1], c[2], radius * rng.uniform(0.5, 1.5), mat)
    return body


def tris(rng, n, centre, spread, size, flat=False):
    pts, faces = [], []
    for i in range(n):
        c = centre + rng.uniform(-spread, spread, 3)
        for _ in range(3):
            p = c + rng.uniform(-size, size, 3)
            if flat:
                p[2] = centre[2]
            pts.append("(%.9g,%.9g,%.9g)" % tuple(p))
        faces.append("(%d,%d,%d)" % (3 * i, 3 * i + 1, 3 * i + 2))
    return "polymesh { material = { diffuse = (0.6,0.7,0.8); emissive = (0.02,0.02,0.02); }; points = (%s); faces = (%s); }\n" % (",".join(pts), ",".join(faces))


def cases():
    rng = np.random.default_rng(77)
    z = np.zeros(3)
    ground = lambda s: "translate(0,-1,0, rotate(1,0,0,-1.5707963, scale(%g, square { material = { diffuse = (0.5,0.5,0.5); } })))\n" % s
    yield "tiny objects in a 4000-unit scene", H + cam((0, 1, 6)) + ground(4000) + spheres(rng, 60, z, 1.5, 0.15) + tris(rng, 200, z, 1.5, 0.05), z
    yield "fine mesh: triangles a few code steps wide", H + cam((0, 0.5, 5)) + ground(40) + tris(rng, 3000, z, 1.0, 0.004), z
    # (the reference's own slab test, bbox.cu:21-33, stops being reliable once the rounding of the
    # coordinates, 6e-8 x their magnitude, reaches the 1e-5 x diagonal padding of the boxes: the oracle's
    # tree walk then differs from its test of every primitive.  These two stay inside that envelope.)
    off = np.array([60.0, -40.0, 100.0])
    yield "scene 120 units from the origin", H + cam(off + np.array([0, 1, 6]), off) + spheres(rng, 40, off, 1.5, 0.3) + tris(rng, 100, off, 1.5, 0.2), off
    yield "flat scene (zero z extent)", H + cam((0.3, 0.2, 4)) + tris(rng, 300, z, 1.5, 0.2, flat=True), z
    yield "camera 15 scene extents away", H + cam((0, 15, 60), fov=4) + spheres(rng, 40, z, 1.2, 0.3) + tris(rng, 100, z, 1.2, 0.2), z
    yield ("two distant clusters", H + cam((0, 1, 9)) + spheres(rng, 30, np.array([-400.0, 0, 0]), 1.0, 0.3) +
           spheres(rng, 30, np.array([400.0, 0, -3]), 1.0, 0.3) + tris(rng, 100, z, 1.5, 0.3), z)


@pytest.mark.parametrize("name,text,centre", list(cases()), ids=[c[0] for c in cases()])
def test_quantised_nodes_keep_every_hit(ctx, name, text, centre):
    sc = P.Scene.from_text(text)
    rng = np.random.default_rng(5)
    s = sc.soa.contents
    eye = np.array([s.cam_eye[0], s.cam_eye[1], s.cam_eye[2]], np.float32)
    for build in (False, True):
        if build:
            ctx.upload_build(sc)
        else:
            ctx.upload(sc)
        p = P.Context.params(96, 74, 2, 4, seed=9)
        mean, rgb, st = ctx.render(p)
        omean, orgb, c = O.render(sc, p)
        assert st.rays_cast == c.rays and same(mean, omean) and np.array_equal(rgb, orgb), (name, build)
        # probes: camera-like rays, rays from inside the scene, axis-parallel rays
        n = 6000
        o = np.where(rng.integers(0, 2, (n, 1)) == 0, eye[None, :] + rng.normal(size=(n, 3)).astype(np.float32) * 0.01,
                     (centre[None, :] + rng.uniform(-2, 2, (n, 3))).astype(np.float32)).astype(np.float32)
        look = np.array([s.cam_look[0], s.cam_look[1], s.cam_look[2]], np.float32)
        d = (look[None, :] + rng.normal(size=(n, 3)).astype(np.float32) * 0.2).astype(np.float32)
        d[n // 2:] = rng.normal(size=(n - n // 2, 3)).astype(np.float32)
        d[:200, 0] = 0
        d[200:400, 1] = 0
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        rays = np.concatenate([o, d], 1).astype(np.float32)
        ids, t, nrm = ctx.trace_rays(rays)
        oi, ot, on, _ = O.trace_rays(sc, rays, O.BRUTE)
        assert np.array_equal(ids, oi) and same(t, ot) and same(nrm, on), (name, build)


def test_plane_codes_enclose_the_fp32_boxes(ctx, scenes_small):
    """Structure of the tree the kernels walk (mi_scene_download_nodes4): every quantised box encloses the
    fp32 box it was made from by at most a code step plus the margin, empty seats are inverted on every
    axis, refs are carried over, the grid spans the root's box, and the treelet copies equal their
    originals up to the redirected refs."""
    MI_DONE = 0x7FFFFFFF
    for name, build in (("simple", False), ("cornell", False), ("glass", True), ("blob3", False), ("blob3", True)):
        sc = scenes_small[name]
        if build:
            ctx.upload_build(sc)
        else:
            ctx.upload(sc)
        f, q, root, tb, qs, qb = ctx.download_nodes4()
        n = f.shape[0]
        assert n > 0 and 0 <= root < n and q.shape == (n, 16)
        refs = f[:, 24:28].view(np.int32)
        assert np.array_equal(q[:, 12:16].view(np.int32), refs)
        lo = f[:, 0:12].reshape(n, 3, 4).astype(np.float64)
        hi = f[:, 12:24].reshape(n, 3, 4).astype(np.float64)
        codes = q[:, 0:12].reshape(n, 3, 4)
        F = lambda c: ((np.uint32(0x3F000000) | (c.astype(np.uint32) << np.uint32(8))).view(np.float32)).astype(np.float64)
        qlo = F(codes & 0xFFFF) * qs[None, :, None].astype(np.float64) + qb[None, :, None].astype(np.float64)
        qhi = F(codes >> 16) * qs[None, :, None].astype(np.float64) + qb[None, :, None].astype(np.float64)
        used = (refs != MI_DONE)[:, None, :].repeat(3, 1)
        step = qs.astype(np.float64)[None, :, None] * 2.0 ** -15        # the coarser of the two binades
        assert np.all(qlo[used] <= lo[used]) and np.all(qhi[used] >= hi[used]), name
        assert np.all((lo - qlo)[used] <= (1.3 * step.repeat(n, 0).repeat(4, 2))[used]) and np.all((qhi - hi)[used] <= (1.3 * step.repeat(n, 0).repeat(4, 2))[used]), name
        assert np.all((codes[~used] & 0xFFFF) == 0xFFFF) and np.all((codes[~used] >> 16) == 0), name
        # grid: the root's box maps to F in [0.51, 1.99]
        r = f[root]
        rr = r[24:28].view(np.int32) != MI_DONE
        for a in range(3):
            rlo, rhi = r[4 * a:4 * a + 4][rr].min(), r[12 + 4 * a:16 + 4 * a][rr].max()
            assert abs((rlo - qb[a]) / qs[a] - 0.51) < 1e-4 and abs((rhi - qb[a]) / qs[a] - 1.99) < 1e-4, name
        # treelet copies: boxes equal to some original node's, refs either equal or redirected into the treelet
        assert 0 < tb <= root < n
        base = {f[i, :24].tobytes(): i for i in range(tb)}
        for i in range(tb, n):
            j = base[f[i, :24].tobytes()]
            same_or_copy = (refs[i] == refs[j]) | (refs[i] >= tb)
            assert same_or_copy.all() and np.array_equal(q[i, :12], q[j, :12]), name
