# Wave iterations of the slowest wave vs time of a tiny launch (instrumented kernel).
import sys; sys.path.insert(0,'.')
import pathtracer_amd as P
from pathtracer_amd import scenes as S
sc=S.blob_scene(6)
ctx=P.Context(0); ctx.upload(sc)
for width in (16, 128, 512):
    w,h=sc.image_size(width)
    for cw in (False, True):
        p=P.Context.params(w,h,1,0,count_work=cw)
        print("==", w, h, "count", cw, file=sys.stderr); sys.stderr.flush()
        for i in range(2):
            st=ctx.render_device(p,None,None)
        print("nodes", st.node_visits, "prims", st.prim_tests, file=sys.stderr)
