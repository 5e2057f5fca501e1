import os, sys; sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import numpy as np
import pathtracer_amd as P
from pathtracer_amd import scenes as S
os.environ["MI_TRACER_TAIL"] = "0"
sc = P.Scene.from_text(S.cornell_ray()); w, h = sc.image_size(256)
def ctx(**env):
    for k, v in env.items(): os.environ[k] = str(v)
    c = P.Context(0); c.upload(sc)
    for k in env: del os.environ[k]
    return c
a = ctx(MI_TRACER_STRAGGLERS=0)
for name, env in (("s16", dict(MI_TRACER_STRAGGLERS=16)), ("s1", dict(MI_TRACER_STRAGGLERS=1)), ("s64", dict(MI_TRACER_STRAGGLERS=64)),
                  ("s16 one stream", dict(MI_TRACER_STRAGGLERS=16, MI_TRACER_STREAMS=1)), ("s16 no treelet", dict(MI_TRACER_STRAGGLERS=16, MI_TRACER_NO_TREELET=1))):
    b = ctx(**env)
    for spp in (1, 2, 4):
        p = P.Context.params(w, h, spp, 0, seed=5)
        ma, ra, sa = a.render(p); mb, rb, sb = b.render(p)
        d = (ma.view(np.uint32) != mb.view(np.uint32)).any(-1)
        ys, xs = np.nonzero(d)
        print(name, "spp", spp, sa.rays_cast, sb.rays_cast, "pixels differing", int(d.sum()), "rows", (ys.min(), ys.max()) if len(ys) else None, "cols", (xs.min(), xs.max()) if len(xs) else None, flush=True)
