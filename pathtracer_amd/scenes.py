"""Deterministic synthetic scenes for the BASELINE configs.

No scene, mesh or image ships with the reference, and the Stanford
bunny/dragon/Lucy files are not available offline, so every workload here is
generated: C1 ``simple`` and C2 ``cornell`` and C5 ``glass`` as ``.ray`` text (they
exercise the parser), C3/C4 ``blob`` as a procedurally displaced icosphere with the
triangle count of the named model class (81,920 at level 6 standing in for the
69,451-triangle bunny; 1,310,720 at level 8 for the dragon/Lucy class).
"""
import numpy as np


def _f(x):
    # the reference's scalar lexer accepts only [0-9.\-e] (src/parser/Tokenizer.cu:217-225)
    s = f"{float(x):.9g}"
    return s.replace("e+", "e")


def _v(v):
    return "(" + ",".join(_f(x) for x in v) + ")"


def mat_text(diffuse=None, emissive=None, reflective=None, transmissive=None, index=None):
    parts = []
    if emissive is not None:
        parts.append(f"emissive = {_v(emissive)};")
    if diffuse is not None:
        parts.append(f"diffuse = {_v(diffuse)};")
    if reflective is not None:
        parts.append(f"reflective = {_v(reflective)};")
    if transmissive is not None:
        parts.append(f"transmissive = {_v(transmissive)};")
    if index is not None:
        parts.append(f"index = {_f(index)};")
    return "material = { " + " ".join(parts) + " }"


def mat20(diffuse=(0, 0, 0), emissive=(0, 0, 0), reflective=(0, 0, 0), transmissive=(0, 0, 0), index=1.0):
    """20 floats ke,kd,kr,kt,index as the parser would build them (w = 0; index broadcast)."""
    return [*emissive, 0, *diffuse, 0, *reflective, 0, *transmissive, 0, index, index, index, index]


HEADER = "SBT-raytracer 1.0\n"


def simple_ray():
    """C1: three spheres (diffuse, mirror, glass) on a 12x12 square, one point light."""
    return HEADER + f"""
// C1 simple scene
camera {{ position = (0,1,6); viewdir = (0,-0.1,-1); updir = (0,1,0); fov = 45; aspectratio = 1; }}
point_light {{ position = (2,5,3); colour = (1,1,1); }}
translate(-2.1,0.0,0.0, sphere {{ {mat_text(diffuse=(0.8, 0.1, 0.1))} }})
translate(0.0,0.0,-0.5, sphere {{ {mat_text(diffuse=(0.85, 0.85, 0.85), reflective=(0.9, 0.9, 0.9))} }})
translate(2.1,0.0,0.5, scale(0.9, sphere {{ {mat_text(diffuse=(0.05, 0.05, 0.05), transmissive=(0.9, 0.9, 0.9), index=1.5)} }}))
translate(0,-1.0,0, rotate(1,0,0,-1.5707963, scale(12, square {{ {mat_text(diffuse=(0.6, 0.6, 0.55))} }})))
"""


def _quad(a, b, c, d, mat):
    pts = ",".join(_v(p) for p in (a, b, c, d))
    return f"polymesh {{ {mat}; points = ({pts}); faces = ((0,1,2,3)); }}\n"


def cornell_ray():
    """C2: closed Cornell box, emissive ceiling panel (ke = 1 -> x32), two boxes."""
    w = mat_text(diffuse=(0.73, 0.73, 0.73))
    red = mat_text(diffuse=(0.65, 0.05, 0.05))
    green = mat_text(diffuse=(0.12, 0.45, 0.15))
    light = mat_text(emissive=(1, 1, 1), diffuse=(0, 0, 0))
    s = HEADER + "// C2 Cornell box\n"
    s += "camera { position = (0,1,3.4); viewdir = (0,0,-1); updir = (0,1,0); fov = 40; aspectratio = 1; }\n"
    s += _quad((-1, 0, 1), (1, 0, 1), (1, 0, -1), (-1, 0, -1), w)      # floor
    s += _quad((-1, 2, 1), (-1, 2, -1), (1, 2, -1), (1, 2, 1), w)      # ceiling
    s += _quad((-1, 0, -1), (1, 0, -1), (1, 2, -1), (-1, 2, -1), w)    # back
    s += _quad((-1, 0, 1), (-1, 0, -1), (-1, 2, -1), (-1, 2, 1), red)  # left
    s += _quad((1, 0, 1), (1, 2, 1), (1, 2, -1), (1, 0, -1), green)    # right
    s += _quad((-1, 0, 3.5), (-1, 2, 3.5), (1, 2, 3.5), (1, 0, 3.5), w)  # wall behind the camera
    s += _quad((-1, 0, 1), (-1, 2, 1), (-1, 2, 3.5), (-1, 0, 3.5), w)
    s += _quad((1, 0, 1), (1, 0, 3.5), (1, 2, 3.5), (1, 2, 1), w)
    s += _quad((-1, 0, 1), (-1, 0, 3.5), (1, 0, 3.5), (1, 0, 1), w)
    s += _quad((-1, 2, 1), (1, 2, 1), (1, 2, 3.5), (-1, 2, 3.5), w)
    s += _quad((-0.35, 1.995, 0.3), (-0.35, 1.995, -0.3), (0.35, 1.995, -0.3), (0.35, 1.995, 0.3), light)
    s += f"translate(-0.35,0.6,-0.3, rotate(0,1,0,0.3, scale(0.6,1.2,0.6, box {{ {w} }})))\n"
    s += f"translate(0.4,0.3,0.3, rotate(0,1,0,-0.35, scale(0.6,0.6,0.6, box {{ {w} }})))\n"
    return s


def glass_ray(nx=6, ny=3, nz=6):
    """C5: grid of glass spheres + glass boxes over a mirror floor in a closed emissive room."""
    glass = mat_text(diffuse=(0.02, 0.02, 0.02), transmissive=(0.95, 0.95, 0.95), index=1.5)
    mirror = mat_text(diffuse=(0.05, 0.05, 0.05), reflective=(0.9, 0.9, 0.9))
    wall = mat_text(diffuse=(0.7, 0.7, 0.75))
    light = mat_text(emissive=(0.5, 0.5, 0.45), diffuse=(0.2, 0.2, 0.2))
    s = HEADER + "// C5 glass-heavy scene\n"
    s += "camera { position = (0,2.2,7.5); viewdir = (0,-0.18,-1); updir = (0,1,0); fov = 42; aspectratio = 1.7777778; }\n"
    R = 5.0
    s += _quad((-R, 0, R), (R, 0, R), (R, 0, -R), (-R, 0, -R), mirror)
    s += _quad((-R, 5, R), (-R, 5, -R), (R, 5, -R), (R, 5, R), light)
    s += _quad((-R, 0, -R), (R, 0, -R), (R, 5, -R), (-R, 5, -R), wall)
    s += _quad((-R, 0, R), (-R, 0, -R), (-R, 5, -R), (-R, 5, R), wall)
    s += _quad((R, 0, R), (R, 5, R), (R, 5, -R), (R, 0, -R), wall)
    s += _quad((-R, 0, 2 * R), (-R, 5, 2 * R), (R, 5, 2 * R), (R, 0, 2 * R), wall)
    s += _quad((-R, 0, R), (-R, 5, R), (-R, 5, 2 * R), (-R, 0, 2 * R), wall)
    s += _quad((R, 0, R), (R, 0, 2 * R), (R, 5, 2 * R), (R, 5, R), wall)
    s += _quad((-R, 0, R), (-R, 0, 2 * R), (R, 0, 2 * R), (R, 0, R), wall)
    s += _quad((-R, 5, R), (R, 5, R), (R, 5, 2 * R), (-R, 5, 2 * R), light)
    for ix in range(nx):
        for iy in range(ny):
            for iz in range(nz):
                x = (ix - (nx - 1) / 2) * 1.1
                y = 0.45 + iy * 1.0
                z = (iz - (nz - 1) / 2) * 1.1 - 0.5
                s += f"translate({_f(x)},{_f(y)},{_f(z)}, scale(0.4, sphere {{ {glass} }}))\n"
    for k in range(4):
        s += f"translate({_f(-3.6 + 2.4 * k)},0.5,2.6, rotate(0,1,0,{_f(0.2 + 0.3 * k)}, scale(0.7,1.0,0.7, box {{ {glass} }})))\n"
    return s


def icosphere(level):
    """Unit icosphere: (verts float64[n,3], faces int32[m,3]); m = 20 * 4**level."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2],
                  [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11],
                  [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    for _ in range(level):
        n = len(v)
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
        e.sort(axis=1)
        key = e[:, 0] * n + e[:, 1]
        uniq, inv = np.unique(key, return_inverse=True)
        a, b = uniq // n, uniq % n
        mid = v[a] + v[b]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        v = np.concatenate([v, mid])
        m = n + inv.reshape(3, -1).T  # midpoints of edges (01, 12, 20) per face
        f = np.concatenate([np.stack([f[:, 0], m[:, 0], m[:, 2]], 1), np.stack([f[:, 1], m[:, 1], m[:, 0]], 1),
                            np.stack([f[:, 2], m[:, 2], m[:, 1]], 1), m])
    return v, f.astype(np.int32)


def blob_mesh(level):
    """Displaced icosphere, r = 1 + 0.15 sin5x cos4y + 0.08 sin(9z+1)."""
    v, f = icosphere(level)
    r = 1.0 + 0.15 * np.sin(5 * v[:, 0]) * np.cos(4 * v[:, 1]) + 0.08 * np.sin(9 * v[:, 2] + 1.0)
    return (v * r[:, None]).astype(np.float32), f


BLOB_CAMERA = dict(eye=(0.0, 0.6, 4.2), viewdir=(0.0, -0.12, -1.0), updir=(0.0, 1.0, 0.0), fov=38.0, aspect=16.0 / 9.0)


def blob_parts(level):
    """Mesh parts of the C3/C4 scene: the blob plus a ground quad, open to the white sky."""
    v, f = blob_mesh(level)
    ground_v = np.array([[-8, -1.25, 8], [8, -1.25, 8], [8, -1.25, -8], [-8, -1.25, -8]], np.float32)
    ground_f = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    body = mat20(diffuse=(0.72, 0.55, 0.4), reflective=(0.15, 0.15, 0.15))
    ground = mat20(diffuse=(0.5, 0.55, 0.5))
    return (v, f, body), (ground_v, ground_f, ground)


def blob_scene(level):
    """C3 (level 6, 81,922 triangles) / C4 (level 8, 1,310,722 triangles) as a Scene."""
    from .scene import Scene
    (v, f, body), ground = blob_parts(level)
    return Scene.from_mesh(v, f, body, BLOB_CAMERA, extra_meshes=[ground])


def blob_ray(level):
    """The same scene as ``.ray`` text (use small levels: text grows with the mesh)."""
    (v, f, _), (gv, gf, _) = blob_parts(level)
    c = BLOB_CAMERA
    s = HEADER + f"camera {{ position = {_v(c['eye'])}; viewdir = {_v(c['viewdir'])}; updir = {_v(c['updir'])}; " \
                 f"fov = {_f(c['fov'])}; aspectratio = {_f(np.float32(c['aspect']))}; }}\n"
    body = mat_text(diffuse=(0.72, 0.55, 0.4), reflective=(0.15, 0.15, 0.15))
    ground = mat_text(diffuse=(0.5, 0.55, 0.5))
    for verts, faces, m in ((v, f, body), (gv, gf, ground)):
        pts = ",".join(_v(p) for p in verts)
        fs = ",".join("(" + ",".join(str(int(i)) for i in face) + ")" for face in faces)
        s += f"polymesh {{ {m}; points = ({pts}); faces = ({fs}); }}\n"
    return s
