# per-level timings (one stream) of the whole C3 frame and of a 1/8 shard
import sys, os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
os.environ["MI_TRACER_DEBUG_LEVELS"] = "1"
import pathtracer_amd as P
from pathtracer_amd import scenes as S
sc = S.blob_scene(6); w, h = sc.image_size(1920)
ctx = P.Context(0); ctx.upload(sc)
for shard in ((0, 1), (3, 8)):
    sys.stderr.write("=== shard %d of %d\n" % shard); sys.stderr.flush()
    p = P.Context.params(w, h, 4, 5, shard=shard, one_stream=True, launch_times=True)
    for i in range(3):
        sys.stderr.write("--- %d\n" % i); sys.stderr.flush()
        st = ctx.render_device(p, None, None)
    print("total", st.ms_total, file=sys.stderr)
