# C4 (1.3 M triangles, 4K, 256 spp): per-level timings of the last batch
import sys; sys.path.insert(0, __import__("os").environ.get("GRAFT_REPO_ROOT","."))
import pathtracer_amd as P
from pathtracer_amd import scenes as S
sc=S.blob_scene(8); w,h=sc.image_size(3840)
ctx=P.Context(0); ctx.upload(sc)
p=P.Context.params(w,h,16,5)
st=ctx.render_device(p,None,None)
print("total",st.ms_total,"ext",st.ms_extend,"shade",st.ms_shade,"tail",st.ms_tail,"other",st.ms_other,"batches",st.batches,"rays",st.rays_cast, file=sys.stderr)
