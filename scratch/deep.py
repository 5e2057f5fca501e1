# depth 60 at 1080p/16 spp: the default batch shrinks with depth; render completes and equals a small-batch render
import sys, os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
import pathtracer_amd as P
from pathtracer_amd import scenes as S
sc = P.Scene.from_text(S.glass_ray()); w, h = sc.image_size(1920)
ctx = P.Context(0); ctx.upload(sc)
a = torch.zeros(w*h*4, dtype=torch.uint8, device="cuda"); b = torch.zeros_like(a)
st = ctx.render_device(P.Context.params(w, h, 4, 60), None, a.data_ptr())
print("default batch: batches", st.batches, "ms", round(st.ms_total, 1), "rays", st.rays_cast, "GPU mem GB", round(torch.cuda.mem_get_info()[1]/2**30 - torch.cuda.mem_get_info()[0]/2**30, 1))
st2 = ctx.render_device(P.Context.params(w, h, 4, 60, batch_paths=3000000), None, b.data_ptr())
print("3M batch: batches", st2.batches, "ms", round(st2.ms_total, 1), "equal", bool(torch.equal(a, b)), st.rays_cast == st2.rays_cast)
