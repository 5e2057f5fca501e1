import sys; sys.path.insert(0,'.')
import pathtracer_amd as P
from pathtracer_amd import scenes as S
sc=S.blob_scene(6); w,h=sc.image_size(1920)
ctx=P.Context(0); ctx.upload(sc)
prev=0
for d in range(-1,6):
    p=P.Context.params(w,h,4,d)
    st=ctx.render_device(p,None,None)
    st=ctx.render_device(p,None,None)
    print(d, st.rays_cast, st.rays_cast-prev, round(st.ms_total,2), round(st.ms_extend,2), round(st.ms_shade,2), round(st.ms_other,2)); prev=st.rays_cast
