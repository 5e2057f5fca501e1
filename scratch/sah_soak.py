# Soak of the device SAH builder: random scenes (spheres, triangle soups, coincident and degenerate primitives) — the
# device tree must be the host tree node for node, and valid.  usage: python scratch/sah_soak.py [n_scenes]
import os, sys; sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.')); sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'tests'))
import numpy as np
import pathtracer_amd as P
from pathtracer_amd import scenes as S
from test_gpu_bvh_build import device_bounds, check_tree, same_tree, H, CAM
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ctx = P.Context(0)
bad = 0
for seed in range(N):
    r = np.random.default_rng(seed)
    n = int(r.choice([2, 3, 7, 40, 64, 65, 100, 300, 1000, 3000]))
    kind = seed % 5
    body = ""
    pts, faces = [], []
    for i in range(n):
        if kind == 0 or (kind == 4 and i % 2):   # spheres
            x, y, z = r.uniform(-3, 3, 3); s = r.uniform(0.01, 0.5)
            body += f"translate({x:.5f},{y:.5f},{z:.5f}, scale({s:.5f}, sphere {{ material = {{ diffuse = (1,1,1); }} }}))\n"
        else:
            if kind == 2: c = np.array([r.integers(-2, 3), r.integers(-2, 3), r.integers(-2, 3)], float)  # many coincident centroids (lattice)
            elif kind == 3: c = np.array([r.uniform(-3, 3), 0.0, 0.0])                                  # collinear, flat
            else: c = r.uniform(-3, 3, 3)
            e = 0.0 if (kind == 3 and i % 7 == 0) else r.uniform(0.01, 0.6)                             # some zero-area triangles
            a, b = r.normal(size=3) * e, r.normal(size=3) * e
            if kind == 3: a[1] = a[2] = b[1] = b[2] = 0.0
            k = len(pts); pts += [c - (a + b) / 3, c + a - (a + b) / 3, c + b - (a + b) / 3]; faces.append((k, k + 1, k + 2))
    if faces:
        body += "polymesh { points = (" + ",".join(f"({p[0]:.6f},{p[1]:.6f},{p[2]:.6f})" for p in pts) + "); faces = (" + ",".join(f"({a},{b},{c})" for a, b, c in faces) + "); material = { diffuse = (0.5,0.5,0.5); } }\n"
    try:
        sc = P.Scene.from_text(H + CAM + body)
    except Exception as e:
        print(seed, "scene rejected by the loader:", str(e)[:80]); continue
    if sc.n_prims < 2: continue
    ctx.upload(sc)
    nh, rh = ctx.download_nodes(); idh = sc.array("prim_id", (sc.n_prims,), np.int32).copy()
    st = ctx.upload_build(sc)
    nd, rd = ctx.download_nodes(); idd = ctx.download_prim_ids()
    lo, hi, _ = device_bounds(ctx, sc)
    try:
        depth = check_tree(nd, rd, lo, hi)
        ok = same_tree(nh, rh, idh, nd, rd, idd)
    except AssertionError as e:
        depth, ok = -1, False
    if not ok or depth != st.depth:
        bad += 1
        print(f"seed {seed} kind {kind} n {sc.n_prims}: depth {st.depth} checked {depth} host {sc.bvh_depth} same_tree {ok}", flush=True)
print(f"{N} scenes, {bad} differ from the host tree or are invalid")
