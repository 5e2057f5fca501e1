import ctypes as C, numpy as np, sys
sys.path.insert(0, '.')
import pathtracer_amd as P
from pathtracer_amd import scenes
lib = C.CDLL('scratch/bvh8/libsim2.so')
lib.sim2_run.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
sc = scenes.blob_scene(int(sys.argv[1]) if len(sys.argv) > 1 else 6)
s = sc.soa.contents
nodes = sc.array('nodes', (s.n_nodes, 16), np.float32)
tri = np.concatenate([sc.array(n, (s.n_tri, 4), np.float32) for n in ('tri_v0', 'tri_uinv', 'tri_vinv', 'tri_ninv')], axis=1).copy()
w, h = 240, 135
eye = np.array(s.cam_eye[:3], np.float32); u = np.array(s.cam_u[:3], np.float32); v = np.array(s.cam_v[:3], np.float32); look = np.array(s.cam_look[:3], np.float32)
xs, ys = np.meshgrid((np.arange(w) + 0.5) / w - 0.5, (np.arange(h) + 0.5) / h - 0.5)
d = look[None, None] + xs[..., None] * u + ys[..., None] * v
d /= np.linalg.norm(d, axis=-1, keepdims=True)
rays = np.concatenate([np.broadcast_to(eye, d.shape), d], -1).reshape(-1, 6).astype(np.float32).copy()
out = (C.c_double * 9)()
for W, mode, quant, order, trav, what in ((4, 2, 0, 0, 0, "4 DP exact sorted"), (8, 2, 1, 1, 1, "8 DP 8-bit octant-groups"), (6, 2, 1, 1, 1, "6 DP 8-bit octant-groups"),(4, 0, 0, 0, 0, "4-wide rigid exact sorted (today)"), (8, 0, 0, 0, 0, "8 rigid exact sorted"), (8, 1, 0, 0, 0, "8 greedy exact sorted"),
                                          (8, 1, 0, 1, 1, "8 greedy exact octant-groups"), (8, 1, 1, 1, 1, "8 greedy 8-bit octant-groups"), (8, 0, 1, 1, 1, "8 rigid 8-bit octant-groups"),
                                          (4, 0, 1, 1, 1, "4 rigid 8-bit octant-groups")):
    lib.sim2_run(nodes.ctypes.data, s.root_ref, tri.ctypes.data, s.n_tri, rays.ctypes.data, len(rays), W, mode, quant, order, trav, out)
    o = list(out)
    print(f"{what:36s}: nodes {int(o[6]):7d} primary: {o[0]/o[2]:.2f} visits {o[1]/o[2]:.2f} tests {o[7]/o[2]:.2f} steps | bounce: {o[3]/o[5]:.2f} visits {o[4]/o[5]:.2f} tests {o[8]/o[5]:.2f} steps")
