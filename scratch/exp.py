import sys, os; sys.path.insert(0,'.')
import pathtracer_amd as P
from pathtracer_amd import scenes as S
which = sys.argv[1] if len(sys.argv)>1 else "blob"
batch = int(sys.argv[2]) if len(sys.argv)>2 else 0
if which=="blob": sc=S.blob_scene(6); W=1920; s=4; D=5
elif which=="glass": sc=P.Scene.from_text(S.glass_ray()); W=1024; s=4; D=8
elif which=="cornell": sc=P.Scene.from_text(S.cornell_ray()); W=512; s=8; D=5
w,h=sc.image_size(W)
ctx=P.Context(0); ctx.upload(sc)
prev=0; pe=0
out=[]
for d in range(-1,D+1):
    p=P.Context.params(w,h,s,d,batch_paths=batch)
    st=ctx.render_device(p,None,None)
    st=ctx.render_device(p,None,None)
    out.append(f"L{d+1}:{(st.rays_cast-prev)/1e6:.2f}M/{st.ms_extend-pe:.2f}")
    prev=st.rays_cast; pe=st.ms_extend
tag=" ".join(f"{k[10:]}={v}" for k,v in os.environ.items() if k.startswith("MI_TRACER_"))
print(f"{which} b={batch} {tag} total={st.ms_total:.2f} ext={st.ms_extend:.2f} sh={st.ms_shade:.2f} oth={st.ms_other:.2f} Mrays/s={st.rays_cast/st.ms_total/1e3:.0f} | "+" ".join(out))
