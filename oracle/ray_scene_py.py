"""Independent restatement of the reference's `.ray` reader, in Python — TEST INFRASTRUCTURE.

Only tests/ may import this.  It exists because the C++ oracle (oracle/oracle.cpp) consumes the scene that
the product's own loader (pathtracer_amd/host/ray_parser.cpp, scene.cpp) flattened: a wrong transform order,
fan order, cube corner or camera basis there would be invisible to every render-parity test.  This module
reads the same text with code that shares nothing with the product (numpy float32, written from the
reference's sources) and returns the primitives in parse order, so a test can compare the two.

Parity unpinned: the reference ships no scene fixtures; glm (0.9.9.7, not vendored) is restated from its
published column-major conventions.  Comparisons are therefore to a few ulp, not bitwise.

What is restated (reference file:line):
  Parser.cu:19-77     scene level: geometry, camera, lights, `material =` (inherited by what follows)
  Parser.cu:79-130    camera { position, viewdir, updir, fov, aspectratio }
  Parser.cu:132-186   groups `{ ... }`
  Parser.cu:228-317   translate / rotate / scale (one or three factors) / transform (four rows, transposed; each row's
                      fourth value is dropped by parseVec4d, Parser.cu:788-799)
  Parser.cu:319-346   sphere: centre = xform * (0,0,0,1), radius = (xform * (1,0,0,0)).x
  Parser.cu:348-398   box / square -> Trimesh::fromBox / fromSquare (trimesh.cu:131-172): 8 corners, 12 faces
  Parser.cu:452-596   trimesh / polymesh: points, faces (arbitrary fan), materials = ( per-vertex ... )
  Parser.cu:612-660   point_light -> unit sphere, material (colour, 0, 0, 1, index 1)
  Parser.cu:801-895   material { emissive, diffuse, reflective, transmissive, index; ambient/specular/shininess read and dropped }
  transform.h:3-7,37-45  child xform = parent * own; point = (M * (v,1)).xyz
  camera.cu:34-65     setLook(viewDir, upDir), setFOV (pi = 3.141592f), update()
  Tokenizer.cu:217-225  a scalar is a run of [0-9.\\-e]
"""
import re

import numpy as np

F = np.float32


class Material:
    def __init__(self, ke=(0, 0, 0), kd=(0, 0, 0), kr=(0, 0, 0), kt=(0, 0, 0), index=1.0):
        self.ke, self.kd, self.kr, self.kt, self.index = tuple(ke), tuple(kd), tuple(kr), tuple(kt), float(index)

    def copy(self):
        return Material(self.ke, self.kd, self.kr, self.kt, self.index)

    def row(self):
        """ke, kd, kr, kt (xyz each) and index — what shading reads (material.h:108-116)."""
        return np.array(list(self.ke) + list(self.kd) + list(self.kr) + list(self.kt) + [self.index], F)


_TOKEN = re.compile(r"\s+|//[^\n]*|/\*.*?\*/|(?P<num>[0-9.\-][0-9.\-e]*)|(?P<id>[A-Za-z_][A-Za-z0-9_\-]*)|(?P<str>\"[^\"]*\")|(?P<p>[{}()=;,])", re.S)


def _tokens(text):
    out, pos = [], 0
    while pos < len(text):
        m = _TOKEN.match(text, pos)
        if not m:
            raise ValueError(f"bad character {text[pos]!r}")
        pos = m.end()
        if m.lastgroup == "num":
            out.append(("num", F(float(m.group()))))
        elif m.lastgroup == "id":
            out.append(("id", m.group()))
        elif m.lastgroup == "str":
            out.append(("str", m.group()[1:-1]))
        elif m.lastgroup == "p":
            out.append((m.group(), None))
    out.append(("eof", None))
    return out


def _translate(v):
    m = np.eye(4, dtype=F)
    m[:3, 3] = v
    return m


def _scale(v):
    return np.diag(np.array([v[0], v[1], v[2], 1], F))


def _rotate(angle, axis):
    """glm::rotate(angle, axis) of gtx/transform: radians, axis normalised, Rodrigues."""
    a = np.asarray(axis, F)
    a = a / F(np.sqrt(F(a @ a)))
    c, s = F(np.cos(angle)), F(np.sin(angle))
    t = (F(1) - c) * a
    r = np.eye(4, dtype=F)
    # written as math (row, column); glm stores the same matrix column by column
    r[0, 0] = c + t[0] * a[0]; r[1, 0] = t[0] * a[1] + s * a[2]; r[2, 0] = t[0] * a[2] - s * a[1]
    r[0, 1] = t[1] * a[0] - s * a[2]; r[1, 1] = c + t[1] * a[1]; r[2, 1] = t[1] * a[2] + s * a[0]
    r[0, 2] = t[2] * a[0] + s * a[1]; r[1, 2] = t[2] * a[1] - s * a[0]; r[2, 2] = c + t[2] * a[2]
    return r


class RayScene:
    """prims: list in parse order of ("tri", verts float32[3,3], [Material] * 1 or 3) / ("sph", centre[3], radius, Material)."""

    def __init__(self, text):
        self.t, self.i = _tokens(text), 0
        self.prims = []
        self.camera = dict(eye=np.zeros(3, F), viewdir=None, updir=None, fov=None, aspect=None)
        self._scene()

    # ---- token helpers
    def peek(self):
        return self.t[self.i]

    def get(self):
        tok = self.t[self.i]
        self.i += 1
        return tok

    def read(self, kind, word=None):
        k, v = self.get()
        if k != kind or (word is not None and v != word):
            raise ValueError(f"expected {word or kind}, got {k} {v}")
        return v

    def cond(self, kind):
        if self.peek()[0] == kind:
            self.i += 1
            return True
        return False

    def is_id(self, *words):
        k, v = self.peek()
        return k == "id" and v in words

    def num(self):
        return self.read("num")

    def vec3(self):
        self.read("(")
        a = self.num(); self.read(","); b = self.num(); self.read(","); c = self.num()
        self.read(")")
        return np.array([a, b, c], F)

    def vec3_expr(self):
        self.get(); self.read("=")
        v = self.vec3()
        self.cond(";")
        return v

    def scalar_expr(self):
        self.get(); self.read("=")
        v = self.num()
        self.cond(";")
        return v

    # ---- grammar
    GEOM = ("sphere", "box", "square", "cylinder", "cone", "trimesh", "polymesh", "translate", "rotate", "scale", "transform")

    def _scene(self):
        self.read("id", "SBT-raytracer")
        self.num()
        mat, root = Material(), np.eye(4, dtype=F)
        while True:
            k, v = self.peek()
            if k == "eof":
                return
            if k == ";":
                self.get()
            elif k == "{" or self.is_id(*self.GEOM):
                self._element(root, mat)
            elif self.is_id("point_light"):
                self._point_light()
            elif self.is_id("directional_light", "ambient_light"):
                self._skip_block()
            elif self.is_id("camera"):
                self._camera()
            elif self.is_id("material"):
                mat = self._material_expr(mat)
            else:
                raise ValueError(f"unexpected {k} {v}")

    def _skip_block(self):
        self.get(); self.read("{")
        while not self.cond("}"):
            self.get()

    def _camera(self):
        self.get(); self.read("{")
        while not self.cond("}"):
            if self.is_id("position"):
                self.camera["eye"] = self.vec3_expr()
            elif self.is_id("viewdir"):
                self.camera["viewdir"] = self.vec3_expr()
            elif self.is_id("updir"):
                self.camera["updir"] = self.vec3_expr()
            elif self.is_id("fov"):
                self.camera["fov"] = self.scalar_expr()
            elif self.is_id("aspectratio"):
                self.camera["aspect"] = self.scalar_expr()
            else:
                raise ValueError("camera attribute")

    def camera_basis(self):
        """(eye, u, v, look) as Camera::update leaves them (camera.cu:34-65)."""
        c = self.camera
        z, y = -c["viewdir"], c["updir"]
        x = np.cross(y, z).astype(F)
        nh = F(2) * F(np.tan(F(c["fov"]) / (F(180.0) / F(3.141592)) / F(2)))
        return c["eye"], x * nh * F(c["aspect"]), y * nh, -z

    def _element(self, xf, mat):
        if self.peek()[0] == "{":
            self.get()
            while not self.cond("}"):
                self._element(xf, mat)
            return
        word = self.peek()[1]
        if word == "sphere":
            m = self._attrs(mat)
            centre = (xf @ np.array([0, 0, 0, 1], F))[:3]
            radius = (xf @ np.array([1, 0, 0, 0], F))[0]
            self.prims.append(("sph", centre.astype(F), F(radius), m))
        elif word in ("box", "square"):
            m = self._attrs(mat)
            h = F(0.5) if word == "box" else F(0.001)
            corners = [(xf @ np.array([sx * F(0.5), sy * F(0.5), sz * h, 1], F))[:3] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)]
            # corner order p000 p001 p010 p011 p100 p101 p110 p111, faces of trimesh.cu:131-150
            for a, b, c in ((0, 1, 2), (3, 2, 1), (4, 6, 5), (7, 5, 6), (0, 2, 4), (6, 4, 2), (1, 5, 3), (7, 3, 5), (0, 1, 4), (5, 4, 1), (2, 6, 3), (7, 3, 6)):
                self.prims.append(("tri", np.array([corners[a], corners[b], corners[c]], F), [m]))
        elif word in ("cylinder", "cone"):
            self._skip_block()
        elif word in ("trimesh", "polymesh"):
            self._trimesh(xf, mat)
        elif word == "translate":
            self.get(); self.read("(")
            v = [self.num(), (self.read(","), self.num())[1], (self.read(","), self.num())[1]]
            self.read(",")
            self._element((xf @ _translate(v)).astype(F), mat)
            self.read(")"); self.cond(";")
        elif word == "rotate":
            self.get(); self.read("(")
            v = [self.num(), (self.read(","), self.num())[1], (self.read(","), self.num())[1]]
            self.read(","); w = self.num(); self.read(",")
            self._element((xf @ _rotate(w, v)).astype(F), mat)
            self.read(")"); self.cond(";")
        elif word == "scale":
            self.get(); self.read("(")
            x = self.num(); self.read(",")
            if self.peek()[0] == "num":
                y = self.num(); self.read(","); z = self.num(); self.read(",")
            else:
                y = z = x
            self._element((xf @ _scale([x, y, z])).astype(F), mat)
            self.read(")"); self.cond(";")
        elif word == "transform":
            self.get(); self.read("(")
            rows = []
            for _ in range(4):
                self.read("(")
                r = [self.num()]
                for _ in range(3):
                    self.read(","); r.append(self.num())
                self.read(")"); self.read(",")
                r[3] = F(0)  # Parser.cu:788-799: parseVec4d reads four scalars and returns the first three — the fourth
                rows.append(r)  # column of the matrix (the translation, and the 1 of the last row) is zero
            self._element((xf @ np.array(rows, F)).astype(F), mat)
            self.read(")"); self.cond(";")
        else:
            raise ValueError(f"geometry {word}")

    def _attrs(self, mat):
        """sphere / box / square body: material and name only."""
        self.get(); self.read("{")
        m = mat
        while not self.cond("}"):
            if self.is_id("material"):
                m = self._material_expr(mat)
            elif self.is_id("name"):
                self.get(); self.read("="); self.get(); self.cond(";")
            else:
                raise ValueError("attribute")
        return m

    def _trimesh(self, xf, mat):
        self.get(); self.read("{")
        mesh_mat, per_vertex, points, faces = mat, [], [], []
        while not self.cond("}"):
            if self.is_id("material"):
                mesh_mat = self._material_expr(mat)
            elif self.is_id("gennormals"):
                self.get(); self.read(";")
            elif self.is_id("points"):
                self.get(); self.read("="); self.read("(")
                while not self.cond(")"):
                    v = self.vec3()
                    points.append((xf @ np.array([v[0], v[1], v[2], 1], F))[:3].astype(F))
                    self.cond(",")
                self.read(";")
            elif self.is_id("normals"):
                self.get(); self.read("="); self.read("(")
                while not self.cond(")"):
                    self.vec3(); self.cond(",")
                self.read(";")
            elif self.is_id("materials"):
                self.get(); self.read("="); self.read("(")
                while not self.cond(")"):
                    per_vertex.append(self._material(mesh_mat)); self.cond(",")
                self.read(";")
            elif self.is_id("faces"):
                self.get(); self.read("="); self.read("(")
                while not self.cond(")"):
                    self.read("(")
                    idx = [int(self.num())]
                    while self.cond(","):
                        idx.append(int(self.num()))
                    self.read(")"); self.cond(",")
                    a, b = idx[0], idx[1]
                    for c in idx[2:]:  # Parser.cu:588-595: fan around the first vertex
                        faces.append((a, b, c)); b = c
                self.read(";")
            else:
                raise ValueError("trimesh attribute")
        for a, b, c in faces:
            mats = [per_vertex[a], per_vertex[b], per_vertex[c]] if per_vertex else [mesh_mat]
            self.prims.append(("tri", np.array([points[a], points[b], points[c]], F), mats))

    def _point_light(self):
        self.get(); self.read("{")
        pos = col = None
        while not self.cond("}"):
            if self.is_id("position"):
                pos = self.vec3_expr()
            elif self.is_id("color", "colour"):
                col = self.vec3_expr()
            else:
                self.scalar_expr()
        self.prims.append(("sph", pos, F(1), Material(col, (0, 0, 0), (0, 0, 0), (1, 1, 1), 1)))

    def _material_expr(self, parent):
        self.read("id", "material"); self.read("=")
        m = self._material(parent)
        self.cond(";")
        return m

    def _material(self, parent):
        self.read("{")
        m = parent.copy()
        while not self.cond("}"):
            k, word = self.get()
            self.read("=")
            if word in ("emissive", "diffuse", "reflective", "transmissive", "ambient", "specular"):
                v = tuple(float(x) for x in self.vec3())
                self.cond(";")
                if word in ("emissive", "diffuse", "reflective", "transmissive"):
                    setattr(m, {"emissive": "ke", "diffuse": "kd", "reflective": "kr", "transmissive": "kt"}[word], v)
            elif word in ("index", "shininess"):
                v = self.num(); self.cond(";")
                if word == "index":
                    m.index = float(v)
            else:
                raise ValueError(f"material attribute {word}")
        return m
