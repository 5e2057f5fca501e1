"""Host scene objects: ``.ray`` text or procedural meshes -> flattened SoA scene."""
import ctypes as C

import numpy as np

from . import _native as N


class Scene:
    """A loaded scene with its BVH built and flattened (``mi_scene_soa``).

    Counterpart of what ``Tracer::loadScene`` leaves in ``Tracer::scene``
    (reference src/tracer.cu:42-85)."""

    def __init__(self, handle):
        if not handle:
            raise ValueError(N.host.pt_last_error().decode())
        self._h = C.c_void_p(handle)
        self.soa = N.host.pt_scene_soa(self._h)

    @classmethod
    def from_text(cls, text):
        return cls(N.host.pt_scene_load(text.encode(), 0))

    @classmethod
    def from_file(cls, path):
        return cls(N.host.pt_scene_load(str(path).encode(), 1))

    @classmethod
    def from_mesh(cls, verts, faces, material, camera, extra_meshes=(), spheres=()):
        """verts float32[n,3], faces int32[m,3], material = 20 floats (ke,kd,kr,kt,index as
        float4 each), camera = dict(eye, viewdir, updir, fov, aspect).  extra_meshes is a
        list of (verts, faces, material); spheres a list of (center, radius, material)."""
        h = C.c_void_p(N.host.pt_scene_new())

        def add_mat(m):
            arr = (C.c_float * 20)(*[float(x) for x in m])
            return N.host.pt_scene_add_material(h, arr)

        for v, f, m in [(verts, faces, material)] + list(extra_meshes):
            v = np.ascontiguousarray(v, dtype=np.float32)
            f = np.ascontiguousarray(f, dtype=np.int32)
            added = N.host.pt_scene_add_mesh(h, v.ctypes.data, len(v), f.ctypes.data, len(f), add_mat(m))
            if added < 0:
                raise ValueError("face index out of range")
        for c, r, m in spheres:
            N.host.pt_scene_add_sphere(h, (C.c_float * 3)(*c), float(r), add_mat(m))
        f3 = lambda k: (C.c_float * 3)(*[float(x) for x in camera[k]])
        N.host.pt_scene_set_camera(h, f3("eye"), f3("viewdir"), f3("updir"), float(camera["fov"]), float(camera["aspect"]))
        N.host.pt_scene_finalize(h)
        return cls(h.value)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and N is not None and getattr(N, "host", None) is not None:  # module globals may be gone at interpreter exit
            N.host.pt_scene_free(h)

    @property
    def n_prims(self):
        s = self.soa.contents
        return s.n_tri + s.n_sph

    @property
    def aspect(self):
        return N.host.pt_scene_aspect(self._h)

    @property
    def bvh_depth(self):
        return N.host.pt_scene_bvh_depth(self._h)

    @property
    def sah_cost(self):
        return N.host.pt_scene_sah_cost(self._h)

    def image_size(self, width):
        """(width, height) as CommandLineUI::run computes it (src/ui/CommandLineUI.cu:83-84)."""
        import math
        v = width / self.aspect + 0.5
        # std::round: half away from zero
        return width, int(math.floor(v + 0.5)) if v >= 0 else -int(math.floor(-v + 0.5))

    def slot_vertices(self):
        """float32[n_tri, 3, 4] raw triangle vertices in slot order."""
        n = self.soa.contents.n_tri
        p = N.host.pt_scene_slot_vertices(self._h)
        return np.ctypeslib.as_array(p, shape=(n, 3, 4)).copy() if n else np.zeros((0, 3, 4), np.float32)

    def array(self, name, shape, dtype):
        p = getattr(self.soa.contents, name)
        n = int(np.prod(shape))
        if n == 0:
            return np.zeros(shape, dtype)
        return np.ctypeslib.as_array(p, shape=shape).astype(dtype, copy=True)
