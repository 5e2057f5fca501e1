import sys; sys.path.insert(0,'.')
import pathtracer_amd as P
from pathtracer_amd import scenes as S
sc=S.blob_scene(6); w,h=sc.image_size(1920)
ctx=P.Context(0); ctx.upload(sc)
p=P.Context.params(w,h,4,5)
for i in range(3):
    st=ctx.render_device(p,None,None)
print("total",st.ms_total,"ext",st.ms_extend,"shade",st.ms_shade,"tail",st.ms_tail,"other",st.ms_other, file=sys.stderr)
