import sys; sys.path.insert(0,'.')
import pathtracer_amd as P
from pathtracer_amd import scenes as S
sc=S.blob_scene(6); w,h=sc.image_size(1920)
ctx=P.Context(0); ctx.upload(sc)
for n in (1,2,4,8):
    worst=0
    for r in range(n):
        p=P.Context.params(w,h,4,5,shard=(r,n))
        for i in range(3): st=ctx.render_device(p,None,None)
        worst=max(worst,st.ms_total)
        if r==0: s0=st
    print(f"N={n}: slowest shard {worst:.2f} ms (rank0: ext {s0.ms_extend:.2f} shade {s0.ms_shade:.2f} tail {s0.ms_tail:.2f} other {s0.ms_other:.2f}, rays {s0.rays_cast}); ideal {9.4/n:.2f}")
