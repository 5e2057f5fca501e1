import sys; sys.path.insert(0,'.')
import numpy as np
import pathtracer_amd as P
from pathtracer_amd import scenes
from oracle import oracle_py as O
ctx=P.Context(0)
rng = np.random.default_rng(11)
for name,txt in (("glass",scenes.glass_ray()),("blob3",scenes.blob_ray(3)),("cornell",scenes.cornell_ray())):
    sc=P.Scene.from_text(txt); ctx.upload(sc)
    n=20000
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
    occ = ctx.occluded_rays(rays, tmax); ref = O.occluded(sc, rays, tmax)
    bad = np.nonzero(occ != ref)[0]
    print(name, len(bad), "mismatches; gpu sum", occ.sum(), "ref sum", ref.sum())
    ids,t,_ = ctx.trace_rays(rays, want_normals=False)
    for b in bad[:8]:
        print("  ray", b, "gpu", occ[b], "ref", ref[b], "tmax", tmax[b], "closest id", ids[b], "t", t[b])
