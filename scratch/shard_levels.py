import sys, os; sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','.'))
import pathtracer_amd as P
from pathtracer_amd import scenes as S
sc=S.blob_scene(6); w,h=sc.image_size(1920)
ctx=P.Context(0); ctx.upload(sc)
n=int(sys.argv[1]) if len(sys.argv)>1 else 8
p=P.Context.params(w,h,4,5,shard=(0,n))
for i in range(3):
    print("--- render", i, file=sys.stderr, flush=True)
    st=ctx.render_device(p,None,None)
print("N",n,"total",st.ms_total,"ext",st.ms_extend,"shade",st.ms_shade,"tail",st.ms_tail,"other",st.ms_other,"rays",st.rays_cast, file=sys.stderr)
