# Potential of direction-octant binning for continuation rays: the same synthetic level-1 rays (origins = primary hit points
# of C3 in pixel-tile order, directions = random hemisphere samples about +-normal) traced by k_extend in three orders.
import sys, os; sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','.'))
import numpy as np
import pathtracer_amd as P
from pathtracer_amd import scenes as S
sc = S.blob_scene(6); w, h = 1920, 1080
ctx = P.Context(0); ctx.upload(sc)
ids, t, nrm = ctx.trace_primary(w, h, 1)
s = sc.soa.contents
eye, u, v, look = (np.array(x, np.float32)[:3] for x in (s.cam_eye, s.cam_u, s.cam_v, s.cam_look))
py, px = np.divmod(np.arange(w * h), w)
x = (px.astype(np.float32)) / np.float32(w) - np.float32(0.5); y = (py.astype(np.float32)) / np.float32(h) - np.float32(0.5)
d = look[None] + x[:, None] * u[None] + y[:, None] * v[None]; d /= np.linalg.norm(d, axis=1, keepdims=True)
hit = ids >= 0
# pixel-tile order (8x8) like the renderer's slots
tile = (py // 8) * (w // 8) + (px // 8); order = np.lexsort((px % 8, py % 8, tile))
order = order[hit[order]]
o = (eye[None] + (t[order, None] - 1e-5) * d[order]).astype(np.float32)
n = nrm[order]
rng = np.random.default_rng(1)
r = rng.normal(size=n.shape).astype(np.float32); r /= np.linalg.norm(r, axis=1, keepdims=True)
r = np.where((r * n).sum(1, keepdims=True) < 0, -r, r)          # upper hemisphere about the normal
rays = np.concatenate([o, r], 1).astype(np.float32)
rays = np.tile(rays, (4, 1))[:2_000_000] if len(rays) < 2_000_000 else rays[:2_000_000]
octant = (rays[:, 3] < 0) * 1 + (rays[:, 4] < 0) * 2 + (rays[:, 5] < 0) * 4
def blocked(nb):
    idx = np.arange(len(rays)); blk = idx // nb
    return np.lexsort((idx, octant, blk))
for name, perm in (("as is", np.arange(len(rays))), ("octant within 16384", blocked(16384)), ("octant within 262144", blocked(262144)), ("octant, whole stream", np.lexsort((np.arange(len(rays)), octant))),
                   ("random order", rng.permutation(len(rays)))):
    rr = np.ascontiguousarray(rays[perm])
    for rep in range(2):
        i2, t2, _ = ctx.trace_rays(rr, want_normals=False)
    print(name, "hits", int((i2 >= 0).sum()), flush=True)
