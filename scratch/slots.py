import sys, os; sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','.'))
import pathtracer_amd as P
from pathtracer_amd import scenes as S
sc=S.blob_scene(6); w,h=sc.image_size(1920)
ctx=P.Context(0); ctx.upload(sc)
for d in (-1, 0, 1, 5):
    print("== depth", d, flush=True)
    st=ctx.render_device(P.Context.params(w,h,4,d,count_work=True),None,None)
    print("rays", st.rays_cast, "nodes/ray %.3f prims/ray %.3f" % (st.node_visits/st.rays_cast, st.prim_tests/st.rays_cast), flush=True)
