import sys, time, threading; sys.path.insert(0,'.')
import torch
import pathtracer_amd as P
from pathtracer_amd import scenes as S
sc=S.blob_scene(6); w,h=sc.image_size(1920)
ctxs=[P.Context(0) for _ in range(2)]
for c in ctxs: c.upload(sc)
full=P.Context.params(w,h,4,5)
halves=[P.Context.params(w,h,4,5,shard=(r,2)) for r in range(2)]
def run_full():
    ctxs[0].render_device(full,None,None)
def run_half(r, out):
    out[r]=ctxs[r].render_device(halves[r],None,None)
for i in range(3): run_full()
t=time.perf_counter()
for i in range(10): run_full()
print("full, one ctx: %.2f ms"%((time.perf_counter()-t)*100))
out=[None,None]
def both():
    th=[threading.Thread(target=run_half,args=(r,out)) for r in range(2)]
    for x in th: x.start()
    for x in th: x.join()
for i in range(3): both()
t=time.perf_counter()
for i in range(10): both()
print("two halves concurrently: %.2f ms"%((time.perf_counter()-t)*100))
print("half stats: total %.2f ext %.2f shade %.2f tail %.2f"%(out[0].ms_total,out[0].ms_extend,out[0].ms_shade,out[0].ms_tail))
t=time.perf_counter()
for i in range(10):
    run_half(0,out); run_half(1,out)
print("two halves sequentially: %.2f ms"%((time.perf_counter()-t)*100))
