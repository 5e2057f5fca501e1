# Lane utilisation of k_extend: lane-steps (node visits + primitive tests) / (64 x wave iterations), instrumented kernel.
import sys; sys.path.insert(0,'.')
import pathtracer_amd as P
from pathtracer_amd import scenes as S
sc=S.blob_scene(6); w,h=sc.image_size(1920)
ctx=P.Context(0); ctx.upload(sc)
for depth in (0, 5):
    p=P.Context.params(w,h,4,depth,count_work=True)
    print("== depth", depth, file=sys.stderr); sys.stderr.flush()
    st=ctx.render_device(p,None,None)
    print("rays", st.rays_cast, "nodes", st.node_visits, "prims", st.prim_tests, file=sys.stderr)
