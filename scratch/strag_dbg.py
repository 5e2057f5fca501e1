import os, sys; sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import numpy as np
import pathtracer_amd as P
from pathtracer_amd import scenes as S
os.environ["MI_TRACER_TAIL"] = "0"
name = sys.argv[1]
sc = P.Scene.from_text(S.glass_ray() if name == 'glass' else S.cornell_ray()); w, h = sc.image_size(360 if name == 'glass' else 256)
os.environ["MI_TRACER_STRAGGLERS"] = "0"; a = P.Context(0); a.upload(sc)
os.environ["MI_TRACER_STRAGGLERS"] = "16"; b = P.Context(0); b.upload(sc)
for depth in (0, 1, 2, 3, 4, 8):
    p = P.Context.params(w, h, 2 if name == 'glass' else 4, depth, seed=5)
    ma, ra, sa = a.render(p); mb, rb, sb = b.render(p)
    print(depth, sa.rays_cast, sb.rays_cast, "pixels differing", int((ma.view(np.uint32) != mb.view(np.uint32)).any(-1).sum()), flush=True)
