# C3 frame through mi_render_device_noalloc (static 64-MiB workspace) against mi_render_device
import sys, os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
import pathtracer_amd as P
from pathtracer_amd import scenes as S
sc = S.blob_scene(6); w, h = sc.image_size(1920)
ctx = P.Context(0); ctx.upload(sc)
p = P.Context.params(w, h, 4, 5)
n = P.shard_pixels(p)
a = torch.zeros((n, 4), dtype=torch.uint8, device="cuda:0"); b = torch.zeros_like(a)
for i in range(3): s0 = ctx.render_device(p, None, a.data_ptr())
for i in range(2): s1 = ctx.render_device_noalloc(p, None, b.data_ptr())
print("mi_render_device %.2f ms, %d batches; noalloc %.2f ms, %d batches; same image: %s" % (s0.ms_total, s0.batches, s1.ms_total, s1.batches, torch.equal(a, b)))
