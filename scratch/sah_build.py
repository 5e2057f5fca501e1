import os, sys; sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import pathtracer_amd as P
from pathtracer_amd import scenes as S
lvl = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sc = S.blob_scene(lvl)
ctx = P.Context(0)
for k in range(3):
    st = ctx.upload_build(sc)
    print(f"n {sc.n_prims} build {st.ms_build:.2f} ms depth {st.depth} sah {st.sah_cost:.3f}", flush=True)
