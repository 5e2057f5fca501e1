# per-level timings of one frame of a bench workload: python scratch/lv.py <workload>
import sys, os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import pathtracer_amd as P
from pathtracer_amd import scenes as S
import bench
build, width, samples, depth = bench.WORKLOADS[sys.argv[1]]
sc = build(S); w, h = sc.image_size(width)
ctx = P.Context(0); ctx.upload(sc)
p = P.Context.params(w, h, samples, depth)
for i in range(2): st = ctx.render_device(p, None, None)
print("total", st.ms_total, "ext", st.ms_extend, "shade", st.ms_shade, "tail", st.ms_tail, "other", st.ms_other, "rays", st.rays_cast, file=sys.stderr)
