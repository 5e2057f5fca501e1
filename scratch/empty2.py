import sys, os; sys.path.insert(0,'.')
import pathtracer_amd as P
from pathtracer_amd import scenes as S
sc=P.Scene.from_text(S.HEADER+"camera { position = (0,0,5); viewdir = (0,0,-1); updir = (0,1,0); fov = 60; aspectratio = 1.7777; }\n")
ctx=P.Context(0); ctx.upload(sc)
p=P.Context.params(1920,1080,4,-1)
st=ctx.render_device(p,None,None)
st=ctx.render_device(p,None,None)
tag=" ".join(f"{k[10:]}={v}" for k,v in os.environ.items() if k.startswith("MI_TRACER_"))
print('empty', tag, st.rays_cast, round(st.ms_total,3), round(st.ms_extend,3), round(st.ms_shade,3), round(st.ms_other,3))
