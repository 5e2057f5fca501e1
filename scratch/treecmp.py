# Host SAH tree against device-built tree: shape of the binary and 4-wide trees, work per ray, time.
# usage: python scratch/treecmp.py C3|C4 host|device   (environment selects builder variants)
import sys, os, time; sys.path.insert(0, '.')
import numpy as np
import pathtracer_amd as P
from pathtracer_amd import scenes as S
import bench

wl = {'C3': 'C3-blob82k-1080p-16spp', 'C4': 'C4-blob1.3M-4k-256spp', 'C5': 'C5-glass-1024-16spp', 'C2': 'C2-cornell-512-64spp'}[sys.argv[1]]
how = sys.argv[2]
build, width, samples, depth = bench.WORKLOADS[wl]
sc = build(S); w, h = sc.image_size(width)
ctx = P.Context(0)
t0 = time.time()
if how == 'host':
    ctx.upload(sc)
else:
    ctx.upload_build(sc)
t_up = time.time() - t0
nodes, root = ctx.download_nodes()
n2 = len(nodes)
nd = nodes
refs = nd[:, 12:14].view(np.int32)
# depth of every leaf (iterative)
depth_of = np.zeros(n2, np.int32)
order = [root]; depth_of[root] = 1
leaf_depths = []
i = 0
while i < len(order):
    n = order[i]; i += 1
    for c in range(2):
        r = refs[n, c]
        if r >= 0:
            depth_of[r] = depth_of[n] + 1; order.append(r)
        else:
            leaf_depths.append(depth_of[n] + 1)
leaf_depths = np.array(leaf_depths)
n4a, q4, root4, tbase, qs_, qb_ = ctx.download_nodes4()
n4 = n4a.reshape(-1, 32)[:tbase]
r4 = n4[:, 24:28].view(np.int32)
filled = (r4 != 0x7fffffff).sum(1)
leaves4 = ((r4 < 0)).sum(1)
print(f"{wl} {how}: upload {t_up*1e3:.0f} ms, sah {ctx.scene_quality():.3f}, binary nodes {n2}, leaf depth mean {leaf_depths.mean():.2f} max {leaf_depths.max()}, "
      f"4-wide nodes {len(n4)}, seats filled {filled.mean():.3f} (hist {np.bincount(filled, minlength=5).tolist()}), leaf seats/node {leaves4.mean():.3f}")
p = P.Context.params(w, h, samples, depth, count_work=True)
st = ctx.render_device(p, None, None)
print(f"  nodes/ray {st.node_visits / st.rays_cast:.3f} prims/ray {st.prim_tests / st.rays_cast:.3f}")
p = P.Context.params(w, h, samples, depth)
best = 1e9
for k in range(6):
    st = ctx.render_device(p, None, None)
    best = min(best, st.ms_total)
print(f"  frame {best:.3f} ms  {st.rays_cast / best / 1e3:.0f} Mrays/s")
