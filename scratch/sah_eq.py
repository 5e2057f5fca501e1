# The device top-down builder against the host builder: same tree?  usage: python scratch/sah_eq.py
import os, sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import pathtracer_amd as P
from pathtracer_amd import scenes as S
from test_gpu_bvh_build import device_bounds, check_tree

def lockstep(nh, rh, idh, nd, rd, idd):
    """number of interior nodes compared, first difference (or None)"""
    refs_h = nh[:, 12:14].view(np.int32); refs_d = nd[:, 12:14].view(np.int32)
    st = [(rh, rd)]; cnt = 0
    while st:
        a, b = st.pop(); cnt += 1
        if not np.array_equal(nh[a, :12].view(np.uint32), nd[b, :12].view(np.uint32)):
            d = np.abs(nh[a, :12] - nd[b, :12]).max()
            if d > 0: return cnt, f"boxes differ at host node {a} / device node {b} by {d}"
        for c in range(2):
            x, y = int(refs_h[a, c]), int(refs_d[b, c])
            if (x < 0) != (y < 0): return cnt, f"leaf/interior mismatch at {a}/{b} child {c}"
            if x < 0:
                if idh[~x] != idd[~y]: return cnt, f"leaf primitive {idh[~x]} vs {idd[~y]} at {a}/{b}"
            else: st.append((x, y))
    return cnt, None

H = S.HEADER
CAM = "camera { position = (0,0,5); viewdir = (0,0,-1); updir = (0,1,0); fov = 60; aspectratio = 1; }\n"
def spheres(n, seed):
    r = np.random.default_rng(seed)
    body = ""
    for i in range(n):
        x, y, z = r.uniform(-3, 3, 3); s = r.uniform(0.05, 0.4)
        body += f"translate({x:.5f},{y:.5f},{z:.5f}, scale({s:.5f}, sphere {{ material = {{ diffuse = (1,1,1); }} }}))\n"
    return P.Scene.from_text(H + CAM + body)
tri = "polymesh { points = ((-1,-1,0),(1,-1,0),(0,1,0)); faces = ((0,1,2)); material = { diffuse = (0.5,0.5,0.5); } }\n"
cases = [(f"spheres{n}", (lambda n=n: spheres(n, n))) for n in (2, 3, 5, 63, 64, 65, 66, 129, 200, 1000, 5000)]
cases += [("coincident100", lambda: P.Scene.from_text(H + CAM + tri * 100)), ("coincident300", lambda: P.Scene.from_text(H + CAM + tri * 300))]
cases += [("cornell", lambda: P.Scene.from_text(S.cornell_ray())), ("glass", lambda: P.Scene.from_text(S.glass_ray())), ("blob3", lambda: S.blob_scene(3)),
          ("blob5", lambda: S.blob_scene(5)), ("blob6", lambda: S.blob_scene(6))]
if len(sys.argv) > 1: cases.append(("blob8", lambda: S.blob_scene(8)))
ctx = P.Context(0)
os.environ["MI_TRACER_BUILDER"] = "sah"
for name, make in cases:
    sc = make()
    ctx.upload(sc)
    nh, rh = ctx.download_nodes(); idh = sc.array("prim_id", (sc.n_prims,), np.int32).copy(); qh = ctx.scene_quality()
    st = ctx.upload_build(sc)
    st = ctx.upload_build(sc)
    nd, rd = ctx.download_nodes(); idd = ctx.download_prim_ids(); qd = ctx.scene_quality()
    lo, hi, _ = device_bounds(ctx, sc)
    depth = check_tree(nd, rd, lo, hi) if sc.n_prims < 200000 else -1
    cnt, diff = lockstep(nh, rh, idh, nd, rd, idd)
    print(f"{name}: n {sc.n_prims} build {st.ms_build:.2f} ms depth {st.depth} (checked {depth}) sah host {qh:.4f} device {qd:.4f}; lockstep {cnt}/{len(nh)} nodes: {diff or 'IDENTICAL'}", flush=True)
