"""GPU edge cases through the C-ABI: empty / one-primitive scenes, ragged image sizes,
multi-batch renders, shards, bounce-ray probes incl. axis-parallel rays, per-vertex
materials, a BVH deeper than the LDS stack, and the Tracer-shaped interface."""
import ctypes as C

import numpy as np
import pytest

import pathtracer_amd as P
from pathtracer_amd import _native as N
from pathtracer_amd import scenes, sharding
from oracle import oracle_py as O

pytestmark = pytest.mark.gpu
H = scenes.HEADER
CAM = "camera { position = (0,0,5); viewdir = (0,0,-1); updir = (0,1,0); fov = 60; aspectratio = 1; }\n"


def same(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def test_empty_scene_is_white(ctx):
    sc = P.Scene.from_text(H + CAM)
    ctx.upload(sc)
    mean, rgb, st = ctx.render(P.Context.params(33, 17, 2, 3))
    assert np.all(mean == 1.0) and np.all(rgb == O.tonemap_byte(1.0)) and st.rays_cast == 33 * 17 * 4
    ids, t, _ = ctx.trace_primary(33, 17, 1)
    assert np.all(ids == -1) and np.all(t == np.float32(12345678.9))


@pytest.mark.parametrize("prim", ["sphere { material = { diffuse = (0.7,0.7,0.7); emissive = (0.01,0,0); } }",
                                  "polymesh { material = { diffuse = (0.5,0.9,0.5); }; points = ((-1,-1,0),(1,-1,0),(0,1,0)); faces = ((0,1,2)); }"])
def test_single_primitive_scene(ctx, prim):
    sc = P.Scene.from_text(H + CAM + prim + "\n")
    assert sc.soa.contents.root_ref == -1
    ctx.upload(sc)
    p = P.Context.params(40, 41, 2, 4, seed=5)
    mean, rgb, st = ctx.render(p)
    omean, orgb, cnt = O.render(sc, p)
    assert st.rays_cast == cnt.rays and same(mean, omean) and np.array_equal(rgb, orgb)


def test_ragged_sizes_and_multiple_batches_equal_one_batch(ctx, scenes_small):
    sc = scenes_small["simple"]
    ctx.upload(sc)
    for (w, h) in [(1, 1), (7, 3), (65, 33), (100, 101)]:
        p1 = P.Context.params(w, h, 3, 2, seed=2)
        pb = P.Context.params(w, h, 3, 2, seed=2, batch_paths=1000)     # many small batches
        pt = P.Context.params(w, h, 3, 2, seed=2, tile=(10, 6))         # tiles not a multiple of 8
        m1, r1, s1 = ctx.render(p1)
        mb, rb, sb = ctx.render(pb)
        mt, rt, st = ctx.render(pt)
        om, orr, cnt = O.render(sc, p1)
        assert same(m1, om) and np.array_equal(r1, orr) and s1.rays_cast == cnt.rays
        assert same(mb, om) and np.array_equal(rb, orr) and sb.rays_cast == cnt.rays and (sb.batches > 1 or w * h * 9 <= 1000)
        assert same(mt, om) and np.array_equal(rt, orr) and st.rays_cast == cnt.rays


@pytest.mark.parametrize("count", [2, 3, 8])
def test_shards_compose_to_the_unsharded_image(ctx, scenes_small, count):
    sc = scenes_small["cornell"]
    ctx.upload(sc)
    w, h = sc.image_size(80)
    full_mean, full_rgb, full_st = ctx.render(P.Context.params(w, h, 2, 3, seed=6))
    mean = np.full((h, w, 4), -1, np.float32)
    rgb = np.full((h, w, 3), 7, np.uint8)
    rays = 0
    for r in range(count):
        p = P.Context.params(w, h, 2, 3, seed=6, shard=(r, count))
        st = N.Stats()
        N.check(N.mi.mi_render(ctx._h, C.byref(p), mean.ctypes.data, rgb.ctypes.data, C.byref(st)))
        rays += st.rays_cast
    assert same(mean, full_mean) and np.array_equal(rgb, full_rgb) and rays == full_st.rays_cast


def test_device_outputs_gather_and_assemble(ctx, scenes_small):
    import torch
    sc = scenes_small["glass"]
    ctx.upload(sc)
    w, h = sc.image_size(96)
    count = 3
    full = ctx.render(P.Context.params(w, h, 1, 4, seed=8))[1]
    stride = P.shard_pixels(P.Context.params(w, h, 1, 4, shard=(0, count)))
    assert stride == sharding.shard_pixels(w, h, 0, count)
    gathered = torch.zeros(count, stride * 4, dtype=torch.uint8, device="cuda")
    means = []
    for r in range(count):
        p = P.Context.params(w, h, 1, 4, seed=8, shard=(r, count))
        m = torch.zeros(stride, 4, dtype=torch.float32, device="cuda")
        ctx.render_device(p, m.data_ptr(), gathered[r].data_ptr(), torch.cuda.current_stream().cuda_stream)
        means.append(m.cpu().numpy())
    torch.cuda.synchronize()
    frame = ctx.assemble_rgb8(P.Context.params(w, h, 1, 4, shard=(0, count)), gathered.data_ptr(), stride)
    assert np.array_equal(frame, full)
    dframe = torch.full((h * w * 3,), 77, dtype=torch.uint8, device="cuda")   # every byte must be overwritten
    ctx.assemble_rgb8_device(P.Context.params(w, h, 1, 4, shard=(0, count)), gathered.data_ptr(), stride, dframe.data_ptr(),
                             torch.cuda.current_stream().cuda_stream)
    assert np.array_equal(dframe.cpu().numpy().reshape(h, w, 3), full)
    # the documented shard-local layout matches the host-side mirror in sharding.py
    local = gathered.cpu().numpy().reshape(count, stride, 4)
    assert np.array_equal(sharding.assemble([local[r, :, :3] for r in range(count)], w, h), full)


def test_bounce_ray_probe_matches_brute_force(ctx, scenes_small):
    rng = np.random.default_rng(3)
    for name in ("glass", "blob3", "cornell"):
        sc = scenes_small[name]
        ctx.upload(sc)
        o = rng.uniform(-3, 3, (20000, 3)).astype(np.float32)
        d = rng.normal(size=(20000, 3)).astype(np.float32)
        d[:300, 0] = 0      # 1/d = inf: the slab test runs on the reciprocal clamped to +-1e18
        d[300:600, 1] = 0
        d[600:700, 2] = 0
        d[700:720, :2] = 0
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        # rays along the axes (two infinite reciprocals, -0 among them) from origins on a coarse
        # grid, where they coincide with the axis-aligned faces of the scenes
        d[720:1320] = np.repeat(np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float32), 100, 0)
        d[720:1320][d[720:1320] == 0] = np.where(rng.integers(0, 2, int((d[720:1320] == 0).sum())) == 0, np.float32(0.0), np.float32(-0.0))
        o[720:1320] = np.round(o[720:1320] * 2) / 2
        rays = np.concatenate([o, d], 1)
        ids, t, nrm = ctx.trace_rays(rays)
        oi, ot, on, _ = O.trace_rays(sc, rays, O.BRUTE)
        # a ray that starts exactly ON several primitives has several hits at t = 0; a tree walk stops at
        # the first of them (bbox.cu:33 wants tMax > 0), the test of every primitive keeps the highest id
        ok = ot != 0
        assert ok.sum() > 19000
        assert np.array_equal(ids[ok], oi[ok]) and same(t[ok], ot[ok]) and same(nrm[ok], on[ok])
        assert np.array_equal(t[~ok], ot[~ok])


def test_any_hit_matches_test_of_every_primitive(ctx, scenes_small):
    """mi_occluded_rays (no counterpart in the reference) against the oracle's any-hit
    over every primitive; and consistent with the closest-hit probe."""
    rng = np.random.default_rng(11)
    for name in ("glass", "blob3", "cornell", "simple"):
        sc = scenes_small[name]
        ctx.upload(sc)
        n = 20000
        o = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
        d = rng.normal(size=(n, 3)).astype(np.float32)
        d[:200, 0] = 0
        d[200:400, 1] = 0
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        rays = np.concatenate([o, d], 1)
        tmax = rng.uniform(0.0, 6.0, n).astype(np.float32)
        tmax[:50] = 0.0
        tmax[50:100] = -1.0
        tmax[100:150] = np.float32(12345678.9)
        tmax[150:200] = np.inf
        occ = ctx.occluded_rays(rays, tmax)
        assert np.array_equal(occ, O.occluded(sc, rays, tmax))
        # triangles-only scenes: occluded <=> the closest hit is within tmax (t <= tmax, trimesh.cu:96)
        if sc.soa.contents.n_sph == 0:
            ids, t, _ = ctx.trace_rays(rays, want_normals=False)
            assert np.array_equal(occ.astype(bool), (ids >= 0) & (t <= tmax))
        assert 0 < occ.sum() < n


def test_per_vertex_materials(ctx):
    txt = H + CAM + """polymesh { points = ((-2,-2,0),(2,-2,0),(2,2,0),(-2,2,0)); faces = ((0,1,2,3));
      materials = ({ diffuse = (1,0,0); emissive = (0.02,0,0); }, { diffuse = (0,1,0); reflective = (0.5,0.5,0.5); },
                   { diffuse = (0,0,1); transmissive = (0.6,0.6,0.6); index = 1.4; }, { diffuse = (1,1,0); }); }
    translate(0,0,-3, scale(3, sphere { material = { diffuse = (0.8,0.8,0.8); } }))\n"""
    sc = P.Scene.from_text(txt)
    ctx.upload(sc)
    p = P.Context.params(64, 65, 2, 4, seed=12)
    mean, rgb, st = ctx.render(p)
    omean, orgb, cnt = O.render(sc, p)
    assert st.rays_cast == cnt.rays and same(mean, omean) and np.array_equal(rgb, orgb)


def _chain_scene(levels):
    """Hand-built flattened scene whose BVH forces a traversal stack of `levels` entries:
    node i = (pair node at z = -i-1 [far], node i+1 [near]); the ray runs down -z from z = +levels."""
    n_tri = 2 * levels + 2
    verts = np.zeros((n_tri, 3, 4), np.float32)
    for k in range(n_tri):
        z = float(k // 2)           # pairs (2i, 2i+1) at z = i; larger i = nearer to the ray origin
        verts[k, :, :3] = [[-1, -1, z], [1, -1, z], [0, 1 + 0.01 * (k % 2), z]]
    arrs = {"normal": [], "uinv": [], "vinv": [], "ninv": []}
    for k in range(n_tri):
        ok, (n_, u_, v_, w_) = O.tri_precompute(verts[k])
        assert ok
        for key, val in zip(("normal", "uinv", "vinv", "ninv"), (n_, u_, v_, w_)):
            arrs[key].append(val)
    # nodes: chain nodes 0..levels-1, pair nodes levels..2*levels ; last chain node holds two pair nodes
    n_nodes = n_tri - 1
    nodes = np.zeros((n_nodes, 16), np.float32)
    refs = np.zeros((n_nodes, 2), np.int32)

    def box(zlo, zhi):
        return [-1.1, -1.1, zlo - 0.1, 1.1, 1.2, zhi + 0.1]
    n_chain = levels
    for i in range(n_chain):
        pair_node = n_chain + i
        nodes[i, 0:6] = box(i, i)                       # left: pair i (far)
        nodes[i, 6:12] = box(i + 1, levels)             # right: everything nearer
        refs[i] = [pair_node, i + 1 if i + 1 < n_chain else n_chain + levels]
    for i in range(levels + 1):
        nd = n_chain + i
        nodes[nd, 0:6] = box(i, i)
        nodes[nd, 6:12] = box(i, i)
        refs[nd] = [~(2 * i), ~(2 * i + 1)]
    nodes[:, 12:14] = refs.view(np.float32)
    keep = dict(nodes=np.ascontiguousarray(nodes), v0=np.ascontiguousarray(verts[:, 0]),
                uinv=np.array(arrs["uinv"], np.float32), vinv=np.array(arrs["vinv"], np.float32),
                ninv=np.array(arrs["ninv"], np.float32), normal=np.array(arrs["normal"], np.float32),
                tri_mat=np.zeros((n_tri, 3), np.uint32), has=np.zeros(n_tri, np.uint8),
                mats=np.array(scenes.mat20(diffuse=(0.5, 0.5, 0.5)), np.float32), prim_id=np.arange(n_tri, dtype=np.int32))
    s = N.SceneSoA()
    fp = lambda a: a.ctypes.data_as(N.c_float_p)
    s.n_nodes, s.nodes, s.root_ref = n_nodes, fp(keep["nodes"]), 0
    s.n_tri, s.tri_v0, s.tri_uinv, s.tri_vinv = n_tri, fp(keep["v0"]), fp(keep["uinv"]), fp(keep["vinv"])
    s.tri_ninv, s.tri_normal = fp(keep["ninv"]), fp(keep["normal"])
    s.tri_mat, s.tri_has_vmat = keep["tri_mat"].ctypes.data_as(N.c_u32_p), keep["has"].ctypes.data_as(N.c_u8_p)
    s.n_sph, s.n_mat, s.materials = 0, 1, fp(keep["mats"])
    s.prim_id = keep["prim_id"].ctypes.data_as(N.c_i32_p)
    s.cam_eye[:] = [0, 0, levels + 5.0, 0]
    s.cam_u[:] = [1, 0, 0, 0]
    s.cam_v[:] = [0, 1, 0, 0]
    s.cam_look[:] = [0, 0, -1, 0]
    return s, keep


@pytest.mark.parametrize("levels", [20, 45, 62])
def test_bvh_deeper_than_the_lds_stack(ctx, levels):
    s, keep = _chain_scene(levels)
    N.check(N.mi.mi_scene_upload(ctx._h, C.byref(s)))

    class Fake:  # the oracle binding only needs .soa
        soa = C.pointer(s)
    ids, t, nrm = ctx.trace_primary(24, 24, 1)
    oi, ot, on, cnt = O.trace_primary(Fake, 24, 24, 1, O.BRUTE)
    assert (ids >= 0).sum() > 20 and np.array_equal(ids, oi) and same(t, ot)
    # rays from behind see the pairs in the opposite order: the far pair is pushed at every level
    rays = np.array([[0.01 * k, -0.3, -5, 0, 0, 1] for k in range(64)], np.float32)
    ids, t, _ = ctx.trace_rays(rays)
    oi, ot, _, _ = O.trace_rays(Fake, rays, O.BRUTE)
    assert np.array_equal(ids, oi) and same(t, ot)


def test_too_deep_or_inconsistent_bvh_is_rejected_on_the_host(ctx):
    s, keep = _chain_scene(70)
    assert N.mi.mi_scene_upload(ctx._h, C.byref(s)) == N.MI_ERR_INVALID and b"deeper" in N.mi.mi_last_error()
    s, keep = _chain_scene(5)
    keep["nodes"][2, 12:14] = np.array([999, 1], np.int32).view(np.float32)
    assert N.mi.mi_scene_upload(ctx._h, C.byref(s)) == N.MI_ERR_INVALID
    with pytest.raises(P.MiError):
        ctx.render(P.Context.params(8, 8, 1, 1))      # the failed upload left no scene behind


def test_tracer_interface_matches_the_raw_abi(ctx, scenes_small):
    text = scenes.cornell_ray()
    t = P.HipTracer(depth=3, samples=2, seed=21)
    assert t.load_scene(text=text)
    sc = P.Scene.from_text(text)
    w, h = sc.image_size(72)          # CommandLineUI.cu:83-84 (std::round, half away from zero)
    assert t.aspect_ratio() == sc.aspect
    t.trace_setup(w, h)
    t.trace_image(w, h)
    t.wait_render()
    ctx.upload(sc)
    rgb = ctx.render(P.Context.params(w, h, 2, 3, seed=21))[1]
    assert (w, h) == (72, 73) and np.array_equal(t.get_buffer(), rgb) and t.stats().rays_cast > w * h * 4
    assert not P.HipTracer().load_scene(text="SBT-raytracer 9.9\n")


def test_cli_renders_a_png(tmp_path, scenes_small):
    import os
    import subprocess
    ray = tmp_path / "s.ray"
    ray.write_text(scenes.simple_ray())
    out = tmp_path / "s.png"
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pathtracer_amd", "bin", "ray")
    r = subprocess.run([exe, "-r", "2", "-w", "64", "-s", "2", "-S", "3", str(ray), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    from PIL import Image
    img = np.array(Image.open(out))
    sc = scenes_small["simple"]
    want = O.render(sc, P.Context.params(64, 65, 2, 2, seed=3))[1]
    assert img.shape == (65, 64, 3) and np.array_equal(img[::-1], want)     # PNG rows are flipped
    # -B: BVH built on the GPU; the image is the same, byte for byte
    out2 = tmp_path / "s2.png"
    r = subprocess.run([exe, "-B", "-r", "2", "-w", "64", "-s", "2", "-S", "3", str(ray), str(out2)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert np.array_equal(np.array(Image.open(out2)), img)


def test_paths_deeper_than_the_emission_mask(ctx):
    """The per-path emission mask has 31 bits (fold_history in mi_kernels.h); deeper levels always store their
    emission term.  A mirror box with a glowing sphere keeps paths alive for dozens of levels."""
    mat = "material = { diffuse = (0.99,0.99,0.99); reflective = (0.9,0.9,0.9); }"   # bounce weight = kd (CpuTracer.cu:174)
    walls = ""
    for sx, sy, sz, tx, ty, tz in ((6, 6, 0.1, 0, 0, -3), (6, 6, 0.1, 0, 0, 8), (0.1, 6, 12, -3, 0, 2), (0.1, 6, 12, 3, 0, 2), (6, 0.1, 12, 0, -3, 2), (6, 0.1, 12, 0, 3, 2)):
        walls += "translate(%g,%g,%g, scale(%g,%g,%g, box { %s } ))\n" % (tx, ty, tz, sx, sy, sz, mat)
    text = (H + "camera { position = (0,0,5); viewdir = (0.1,0.05,-1); updir = (0,1,0); fov = 70; aspectratio = 1; }\n" + walls +
            "translate(0.5,-0.4,0, scale(0.6, sphere { material = { diffuse = (0.95,0.95,0.95); emissive = (0.004,0.003,0.002); reflective = (0.5,0.5,0.5); } }))\n")
    sc = P.Scene.from_text(text)
    ctx.upload(sc)
    p = P.Context.params(40, 40, 2, 45, seed=21)
    mean, rgb, st = ctx.render(p)
    omean, orgb, cnt = O.render(sc, p)
    assert cnt.rays > 40 * 40 * 4 * 33, "paths do not reach the levels beyond the mask"
    assert st.rays_cast == cnt.rays and same(mean, omean) and np.array_equal(rgb, orgb)
    pt = P.Context.params(40, 40, 2, 45, seed=21, batch_paths=1500)   # several batches
    mean2, rgb2, st2 = ctx.render(pt)
    assert same(mean2, omean) and st2.rays_cast == cnt.rays
