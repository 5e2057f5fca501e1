# Fixed cost of a level: tiny frames of the C3 scene and of the Cornell box (MI_TRACER_TAIL=0 so every level goes through k_extend/k_shade).
import sys; sys.path.insert(0,'.')
import pathtracer_amd as P
from pathtracer_amd import scenes as S
for name, sc in (("blob", S.blob_scene(6)), ("cornell", P.Scene.from_text(S.cornell_ray()))):
    ctx=P.Context(0); ctx.upload(sc)
    for width in (64, 128, 512):
        w,h=sc.image_size(width)
        p=P.Context.params(w,h,1,5)
        print("==", name, w, h, file=sys.stderr); sys.stderr.flush()
        for i in range(3):
            st=ctx.render_device(p,None,None)
