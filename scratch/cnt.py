import sys, os; sys.path.insert(0,'.')
import pathtracer_amd as P
from pathtracer_amd import scenes as S
sc=S.blob_scene(6); w,h=sc.image_size(1920)
ctx=P.Context(0); ctx.upload(sc)
for d in (-1,0,5):
    p=P.Context.params(w,h,4,d,count_work=True)
    st=ctx.render_device(p,None,None)
    print(d, st.rays_cast, st.node_visits, st.prim_tests, st.ms_extend)
