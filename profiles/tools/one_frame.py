"""One frame of a bench workload through mi_render_device (what rocprofv3 --pmc passes profile).
usage: python profiles/tools/one_frame.py [workload] [host|device]"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (workload table)
import pathtracer_amd as P  # noqa: E402
from pathtracer_amd import scenes as S  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3-blob82k-1080p-16spp"
build, width, samples, depth = bench.WORKLOADS[name]
sc = build(S)
w, h = sc.image_size(width)
ctx = P.Context(0)
if len(sys.argv) > 2 and sys.argv[2] == "device":
    ctx.upload_build(sc)
else:
    ctx.upload(sc)
st = ctx.render_device(P.Context.params(w, h, samples, depth, one_stream=True), None, None)  # one stream: one dispatch per level
print(name, st.ms_total, st.ms_extend, st.rays_cast)
