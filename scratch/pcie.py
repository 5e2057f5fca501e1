import sys, time; sys.path.insert(0,'.')
import numpy as np
import pathtracer_amd as P
from pathtracer_amd import scenes as S
sc=S.blob_scene(6); w,h=sc.image_size(1920)
ctx=P.Context(0); ctx.upload(sc)
p=P.Context.params(w,h,4,5)
for mean in (False, True):
    for i in range(2): ctx.render(p, want_mean=mean)
    t=time.perf_counter(); n=5
    for i in range(n): r=ctx.render(p, want_mean=mean)
    dt=(time.perf_counter()-t)/n*1e3
    st=r[2]
    print("mi_render host outputs (rgb8%s): %.2f ms wall, device part %.2f ms, %.0f Mrays/s PCIe-inclusive"%(" + mean" if mean else "", dt, st.ms_total, st.rays_cast/dt/1e3))
