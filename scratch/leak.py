# context create / upload / build / render / destroy in a loop: device memory must not grow
import sys, os, gc; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
import pathtracer_amd as P
from pathtracer_amd import scenes as S
sc = S.blob_scene(4)
w, h = sc.image_size(320)
free0 = None
for it in range(150):
    ctx = P.Context(0)
    if it % 2: ctx.upload_build(sc)
    else: ctx.upload(sc)
    mean, rgb, st = ctx.render(P.Context.params(w, h, 2, 4, seed=it))
    ids, t, n = ctx.trace_primary(64, 36, 1)
    del ctx; gc.collect()
    if it == 10: free0 = torch.cuda.mem_get_info()[0]
free1 = torch.cuda.mem_get_info()[0]
print("free after 10 iterations %.1f MB, after 150 %.1f MB, growth %.1f MB" % (free0 / 2**20, free1 / 2**20, (free0 - free1) / 2**20))
