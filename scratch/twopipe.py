import sys, time, threading; sys.path.insert(0,'.')
import pathtracer_amd as P
from pathtracer_amd import scenes as S
sc=S.blob_scene(6); w,h=sc.image_size(1920)
ctxs=[P.Context(0) for _ in range(2)]
for c in ctxs: c.upload(sc)
def bench(fn, n=20):
    for i in range(3): fn()
    t=time.perf_counter()
    for i in range(n): fn()
    return (time.perf_counter()-t)/n*1e3
def one(params): return lambda: ctxs[0].render_device(params,None,None)
def two(pa, pb):
    def f():
        th=[threading.Thread(target=lambda c=c,p=p: c.render_device(p,None,None)) for c,p in ((ctxs[0],pa),(ctxs[1],pb))]
        for x in th: x.start()
        for x in th: x.join()
    return f
mk=lambda r,n: P.Context.params(w,h,4,5,shard=(r,n))
print("N=1 full: one ctx %.2f ms | two half-frames concurrently %.2f ms"%(bench(one(mk(0,1))), bench(two(mk(0,2),mk(1,2)))))
print("N=8 shard: one ctx %.2f ms | two halves concurrently %.2f ms"%(bench(one(mk(0,8))), bench(two(mk(0,16),mk(8,16)))))
print("N=2 shard: one ctx %.2f ms | two halves concurrently %.2f ms"%(bench(one(mk(0,2))), bench(two(mk(0,4),mk(2,4)))))
