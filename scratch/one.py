import sys, os; sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','.'))
import pathtracer_amd as P
from pathtracer_amd import scenes as S
sc=S.blob_scene(6); w,h=sc.image_size(1920)
ctx=P.Context(0); ctx.upload(sc)
p=P.Context.params(w,h,4,5)
st=ctx.render_device(p,None,None)
print(st.ms_total, st.ms_extend, st.rays_cast)
