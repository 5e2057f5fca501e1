# per-level timings (one stream) and lane-slot breakdown of one frame: python scratch/lv2.py <workload>
import sys, os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
os.environ["MI_TRACER_DEBUG_LEVELS"] = "1"
import pathtracer_amd as P
from pathtracer_amd import scenes as S
import bench
build, width, samples, depth = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "C3-blob82k-1080p-16spp"]
sc = build(S); w, h = sc.image_size(width)
ctx = P.Context(0); ctx.upload(sc)
p = P.Context.params(w, h, samples, depth, one_stream=True, launch_times=True)
for i in range(3):
    sys.stderr.write("--- frame %d\n" % i); sys.stderr.flush()
    st = ctx.render_device(p, None, None)
print("total", st.ms_total, "ext", st.ms_extend, "shade", st.ms_shade, "tail", st.ms_tail, "rays", st.rays_cast, file=sys.stderr)
sys.stderr.write("--- count_work\n"); sys.stderr.flush()
st = ctx.render_device(P.Context.params(w, h, samples, depth, count_work=True), None, None)
print("nodes/ray", st.node_visits / st.rays_cast, "prims/ray", st.prim_tests / st.rays_cast, file=sys.stderr)
