"""Shared helpers of the GPU parity tests (imported by test modules; pytest puts tests/ on sys.path)."""
import contextlib
import os

import numpy as np

import pathtracer_amd as P
from pathtracer_amd import scenes
from oracle import oracle_py as O


def same(a, b):
    """Raw bits equal (stated tolerance 0)."""
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


@contextlib.contextmanager
def fresh_context(**env):
    """A new mi_ctx created while the given MI_TRACER_* variables are set: the tuning environment is read
    once, in mi_ctx_create, so every setting needs a context of its own (same process, same device)."""
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        c = P.Context(0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    try:
        yield c
    finally:
        c.close()


def random_sphere_field(rng):
    H = scenes.HEADER
    cam = "camera { position = (0,0.5,6); viewdir = (0,-0.05,-1); updir = (0,1,0); fov = 50; aspectratio = %.3f; }\n" % rng.uniform(0.7, 1.9)
    body = ""
    for _ in range(int(rng.integers(1, 40))):
        c = rng.uniform(-2.5, 2.5, 3); r = rng.uniform(0.1, 0.9)
        kind = rng.integers(0, 4)
        mat = ["diffuse = (%.2f,%.2f,%.2f);" % tuple(rng.uniform(0.1, 1, 3)),
               "diffuse = (0.8,0.8,0.8); reflective = (0.9,0.9,0.9);",
               "diffuse = (0.1,0.1,0.1); transmissive = (0.9,0.9,0.9); index = %.2f;" % rng.uniform(1.1, 1.8),
               "diffuse = (0.5,0.5,0.5); emissive = (0.03,0.03,0.02);"][kind]
        body += "translate(%.3f,%.3f,%.3f, scale(%.3f, sphere { material = { %s } }))\n" % (c[0], c[1], c[2], r, mat)
    body += "translate(0,-1.2,0, scale(8, rotate(1,0,0,-1.5708, square { material = { diffuse = (0.6,0.6,0.6); } })))\n"
    return H + cam + body


def fuzz_configurations(ctx, seed, iterations, max_width=180):
    """Random scenes (the generators of pathtracer_amd/scenes.py plus random sphere fields with all four
    material branches), host-built or device-built trees, odd image sizes, tile sizes that are not multiples
    of 8, shards, multi-batch renders, depths -1..8 — each rendered through the C-ABI and compared with the
    oracle bit for bit (ray count, mean radiance, RGB8)."""
    rng = np.random.default_rng(seed)
    makers = [lambda: P.Scene.from_text(scenes.simple_ray()), lambda: P.Scene.from_text(scenes.cornell_ray()),
              lambda: P.Scene.from_text(scenes.glass_ray()), lambda: P.Scene.from_text(scenes.blob_ray(int(rng.integers(1, 5)))),
              lambda: P.Scene.from_text(random_sphere_field(rng))]
    for it in range(iterations):
        sc = makers[int(rng.integers(0, len(makers)))]()
        build = bool(rng.integers(0, 2))
        if build:
            ctx.upload_build(sc)
        else:
            ctx.upload(sc)
        w = int(rng.integers(9, max_width))
        _, h = sc.image_size(w)
        s = int(rng.integers(1, 4))
        d = int(rng.integers(-1, 9))
        rseed = int(rng.integers(0, 2**40))
        tile = [(0, 0), (8, 8), (16, 8), (12, 20), (32, 32), (5, 7)][int(rng.integers(0, 6))]
        cnt = int(rng.integers(1, 5))
        rk = int(rng.integers(0, cnt))
        bp = [0, 64, 1000, 5000, 100000][int(rng.integers(0, 5))]
        p = P.Context.params(w, h, s, d, seed=rseed, tile=tile, shard=(rk, cnt), batch_paths=bp)
        mean, rgb, st = ctx.render(p)
        omean, orgb, c = O.render(sc, p)
        what = dict(it=it, w=w, h=h, s=s, d=d, seed=rseed, tile=tile, shard=(rk, cnt), bp=bp, build=build, prims=sc.n_prims)
        assert st.rays_cast == c.rays, what
        assert same(mean, omean) and np.array_equal(rgb, orgb), what
