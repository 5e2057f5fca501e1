"""BASELINE config C4 on the GPU: the dragon/Lucy-class stand-in (blob level 8 = 1,310,722 primitives),
3840x2160, -s 16 (256 spp), -r 5 — 2.1 G paths, 32 batches per frame, the rotated chunk deal
(deal_rotation), 24-level host tree and ~28-level device-built tree.  The oracle renders two-row bands
(~3.5 M rays each); the full frame is covered by size-independent properties: the host-tree and the
device-tree frames are identical byte for byte, and the ray count is conserved across batch sizes.
Reference path: CpuTracer.cu:99-186; multi-GPU sharding of this config is bench.py's job."""
import numpy as np
import pytest

import pathtracer_amd as P
from pathtracer_amd import scenes
from oracle import oracle_py as O
from gpu_util import same

pytestmark = pytest.mark.gpu

W, S, D = 3840, 16, 5


@pytest.fixture(scope="module")
def blob8():
    return scenes.blob_scene(8)


def test_c4_bands_and_primary_hits_match_the_oracle(ctx, blob8, monkeypatch):
    sc = blob8
    assert sc.n_prims == 1310722
    w, h = sc.image_size(W)
    assert (w, h) == (3840, 2160)
    ctx.upload(sc)
    p = P.Context.params(w, h, S, D, seed=0)
    mean, rgb, st = ctx.render(p)
    assert st.paths == w * h * S * S and st.batches >= 4   # default batch: a quarter of the GPU's memory per stream, at most 256 Mi slots
    for y0 in (3, 400, 771, 1079, 1290, 1500, 1700, 2156):   # eight two-row bands over the frame: ground, silhouette, object, sky
        om, orgb, _ = O.render(sc, p, threads=16, rows=(y0, y0 + 2))
        assert np.array_equal(rgb[y0:y0 + 2], orgb[y0:y0 + 2]), y0
        assert same(mean[y0:y0 + 2], om[y0:y0 + 2]), y0
    # primary hits of a strided set of pixels (1 sample per pixel) against the oracle's tree walk with the canonical
    # tie rule, and a sparser set against its test of every primitive
    pw, ph = 960, 540
    ids, t, nrm = ctx.trace_primary(pw, ph, 1)
    sub = np.arange(0, pw * ph, 37)
    s = sc.soa.contents
    eye, u, v, look = (np.array(x, np.float32)[:3] for x in (s.cam_eye, s.cam_u, s.cam_v, s.cam_look))
    px, py = (sub % pw).astype(np.float32), (sub // pw).astype(np.float32)
    x = (px - np.float32(0.5) + np.float32(0.5)) / np.float32(pw) - np.float32(0.5)
    y = (py - np.float32(0.5) + np.float32(0.5)) / np.float32(ph) - np.float32(0.5)
    d = (look[None] + x[:, None] * u[None]) + y[:, None] * v[None]
    inv = (np.float32(1.0) / np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2], dtype=np.float32)).astype(np.float32)
    d = (d * inv[:, None]).astype(np.float32)
    rays = np.concatenate([np.tile(eye, (len(sub), 1)), d], 1)
    oi, ot, on, _ = O.trace_rays(sc, rays, O.BVH_CANONICAL)
    gi, gt, gn = ctx.trace_rays(rays)
    assert (oi >= 0).sum() > 1000
    assert np.array_equal(gi, oi) and same(gt, ot) and same(gn, on)
    # the probe's own camera rays are the same rays (camera.cu:6-16 restated in numpy above)
    assert np.array_equal(ids[sub], oi) and same(t[sub], ot)
    few = np.arange(0, len(sub), 97)
    bi, bt, _, _ = O.trace_rays(sc, rays[few], O.BRUTE)
    assert np.array_equal(gi[few], bi) and same(gt[few], bt)

    # ray-count conservation and identical frames across batch sizes (the default; 64 Mi slots = 32 batches, round 1's
    # default; 5 M — not a multiple of anything)
    for bp in (64 << 20, 5_000_000):
        _, rgb_b, st_b = ctx.render(P.Context.params(w, h, S, D, seed=0, batch_paths=bp), want_mean=False)
        assert st_b.rays_cast == st.rays_cast and st_b.batches > st.batches
        assert np.array_equal(rgb_b, rgb)

    # C4's defining property — the frame tile-sharded over 2/4/8 GPUs (BASELINE configs[3]): the eight shards (r, 8)
    # rendered one after the other on this GPU into a "gathered" buffer, assembled by mi_assemble_rgb8_device
    # (what rank 0 does after the RCCL gather), must be the single-GPU frame byte for byte, with the rays conserved
    import torch
    n_sh = 8
    stride = P.shard_pixels(P.Context.params(w, h, S, D, seed=0, shard=(0, n_sh)))
    gathered = torch.zeros(n_sh * stride * 4, dtype=torch.uint8, device="cuda:0")
    frame = torch.full((h * w * 3,), 77, dtype=torch.uint8, device="cuda:0")
    sh_rays = sh_paths = 0
    for r in range(n_sh):
        pr = P.Context.params(w, h, S, D, seed=0, shard=(r, n_sh))
        assert P.shard_pixels(pr) <= stride
        st_r = ctx.render_device(pr, None, gathered.data_ptr() + r * stride * 4)
        sh_rays += st_r.rays_cast
        sh_paths += st_r.paths
    ctx.assemble_rgb8_device(P.Context.params(w, h, S, D, seed=0, shard=(0, n_sh)), gathered.data_ptr(), stride, frame.data_ptr())
    torch.cuda.synchronize()
    assert sh_rays == st.rays_cast and sh_paths == st.paths
    assert np.array_equal(frame.cpu().numpy().reshape(h, w, 3), rgb)
    del gathered, frame

    # device-built trees (top-down SAH: the host tree again, built in ~10 ms instead of 3 s; PLOC: another tree, ~28 levels):
    # same frame, same ray count, same primary hits
    for builder in ("sah", "ploc"):
        monkeypatch.setenv("MI_TRACER_BUILDER", builder)
        b = ctx.upload_build(sc)
        assert b.n_nodes == sc.n_prims - 1 and b.depth <= 64 and b.ms_build < 40.0
        assert (b.depth == sc.bvh_depth) == (builder == "sah")
        mean_d, rgb_d, st_d = ctx.render(p)
        assert st_d.rays_cast == st.rays_cast
        assert np.array_equal(rgb_d, rgb) and same(mean_d, mean)
        gi2, gt2, _ = ctx.trace_rays(rays)
        assert np.array_equal(gi2, oi) and same(gt2, ot)
