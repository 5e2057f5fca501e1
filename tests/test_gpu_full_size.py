"""BASELINE-size checks on the GPU (C3: 81,922 primitives, 1920x1080, 16 spp, depth 5; C5: glass, 1024x1025, 16 spp,
depth 8; C1, C2).  The oracle renders each of these WHOLE frames in seconds on the box's 16 host threads (C3: 58.5 M
rays in ~2.3 s), so every pixel's mean radiance and RGB8 bytes and the ray count are compared bit for bit; on top of
that the size-independent properties: determinism, shard composition, ray-count conservation.
Reference path: CpuTracer.cu:99-186."""
import numpy as np
import pytest

import pathtracer_amd as P
from pathtracer_amd import scenes
from oracle import oracle_py as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def blob6():
    return scenes.blob_scene(6)


def test_c3_whole_frame_matches_oracle_bitwise(ctx, blob6):
    ctx.upload(blob6)
    w, h = blob6.image_size(1920)
    assert (w, h) == (1920, 1080)
    p = P.Context.params(w, h, 4, 5, seed=0)
    mean, rgb, st = ctx.render(p)
    om, orgb, cnt = O.render(blob6, p, threads=16)                      # every row of the headline frame
    assert st.rays_cast == cnt.rays and st.paths == w * h * 16
    assert np.array_equal(rgb, orgb)
    assert np.array_equal(mean.view(np.uint32), om.view(np.uint32))
    # determinism + shard composition + ray conservation at full size
    mean2, rgb2, st2 = ctx.render(P.Context.params(w, h, 4, 5, seed=0, batch_paths=3_000_000))
    assert np.array_equal(rgb, rgb2) and st.rays_cast == st2.rays_cast and st.paths == w * h * 16
    acc = np.zeros_like(rgb)
    rays = 0
    import ctypes as C
    from pathtracer_amd import _native as N
    for r in range(4):
        pr = P.Context.params(w, h, 4, 5, seed=0, shard=(r, 4))
        s = N.Stats()
        N.check(N.mi.mi_render(ctx._h, C.byref(pr), None, acc.ctypes.data, C.byref(s)))
        rays += s.rays_cast
    assert np.array_equal(acc, rgb) and rays == st.rays_cast


def test_full_size_primary_hits_match_oracle(ctx, blob6):
    ctx.upload(blob6)
    w, h = 640, 360
    ids, t, nrm = ctx.trace_primary(w, h, 1)
    oi, ot, on, _ = O.trace_primary(blob6, w, h, 1, O.BVH_CANONICAL)   # brute force over 82k prims is too slow here
    agree = (ids == oi).mean()
    assert agree == 1.0, f"primary-hit id agreement {agree}"
    assert np.array_equal(t.view(np.uint32), ot.view(np.uint32)) and np.array_equal(nrm.view(np.uint32), on.view(np.uint32))
    # a brute-force spot check on a sparse subset of the same rays
    sub = np.arange(0, w * h, 997)
    px, py = sub % w, sub // w
    s = blob6.soa.contents
    eye, u, v, look = (np.array(x, np.float32)[:3] for x in (s.cam_eye, s.cam_u, s.cam_v, s.cam_look))
    x = (px.astype(np.float32) - np.float32(0.5) + np.float32(0.5)) / np.float32(w) - np.float32(0.5)
    y = (py.astype(np.float32) - np.float32(0.5) + np.float32(0.5)) / np.float32(h) - np.float32(0.5)
    d = look[None] + x[:, None] * u[None] + y[:, None] * v[None]
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    rays = np.concatenate([np.tile(eye, (len(sub), 1)), d], 1)
    gi, gt, _ = ctx.trace_rays(rays)
    bi, bt, _, _ = O.trace_rays(blob6, rays, O.BRUTE)
    assert np.array_equal(gi, bi) and np.array_equal(gt.view(np.uint32), bt.view(np.uint32))


def test_c5_glass_depth8_whole_frame_matches_oracle_bitwise(ctx):
    sc = P.Scene.from_text(scenes.glass_ray())
    ctx.upload(sc)
    w, h = sc.image_size(1024)
    p = P.Context.params(w, h, 4, 8, seed=0)
    mean, rgb, st = ctx.render(p)
    om, orgb, cnt = O.render(sc, p, threads=16)                          # 16.8 M paths of up to 10 rays
    assert st.rays_cast == cnt.rays and st.paths == w * h * 16
    assert np.array_equal(rgb, orgb)
    assert np.array_equal(mean.view(np.uint32), om.view(np.uint32))


def test_c1_and_c2_whole_frames_match_oracle_bitwise(ctx):
    # BASELINE configs 0 and 1 at their full sizes, every pixel: C1 simple scene 256x257, 1 spp, depth 2 (CpuTracer's own
    # runnable case); C2 Cornell box 512x513, 64 spp, depth 5 (16.8 M paths, ~110 M rays: seconds of oracle time).
    for text, width, s, depth in ((scenes.simple_ray(), 256, 1, 2), (scenes.cornell_ray(), 512, 8, 5)):
        sc = P.Scene.from_text(text)
        w, h = sc.image_size(width)
        ctx.upload(sc)
        p = P.Context.params(w, h, s, depth, seed=0)
        mean, rgb, st = ctx.render(p)
        om, orgb, cnt = O.render(sc, p, threads=16)
        assert (w, h) == (width, width + 1)                     # CommandLineUI.cu:83-84 with aspect 1
        assert st.rays_cast == cnt.rays and st.paths == w * h * s * s
        assert np.array_equal(rgb, orgb)
        assert np.array_equal(mean.view(np.uint32), om.view(np.uint32))
