# Slowest shard of an N-way sharded frame on ONE GPU (every rank's shard rendered in turn): ms of the slowest, mean.
import sys, os; sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','.'))
import bench
import pathtracer_amd as P
from pathtracer_amd import scenes as S
wl = sys.argv[1] if len(sys.argv) > 1 else "C3-blob82k-1080p-16spp"
build, width, samples, depth = bench.WORKLOADS[wl]
sc = build(S); w, h = sc.image_size(width)
ctx = P.Context(0); ctx.upload(sc)
for n in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8").split(",")]:
    ts = []
    for r in range(n):
        p = P.Context.params(w, h, samples, depth, shard=(r, n))
        best = min(ctx.render_device(p, None, None).ms_total for _ in range(4))
        ts.append(best)
    print(f"{wl} N={n}: slowest shard {max(ts):.3f} ms, mean {sum(ts)/n:.3f}, ideal {ts0/n:.3f}" if n > 1 else f"{wl} N=1: {ts[0]:.3f} ms", flush=True)
    if n == 1: ts0 = ts[0]
