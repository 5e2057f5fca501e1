"""ctypes binding of oracle/liboracle.so (the CPU restatement of the reference).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  It reads the flattened scene through the
``mi_scene_soa`` pointer of a ``pathtracer_amd.Scene`` — data layout only.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "liboracle.so")
if not os.path.exists(_PATH):
    raise ImportError(f"{_PATH} missing: run `make -C oracle`")
lib = C.CDLL(_PATH)

BRUTE, BVH_REF_ORDER, BVH_CANONICAL = 0, 1, 2
SINCOS_LIBM, SINCOS_POLY = 0, 1


class Counters(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("node_visits", C.c_uint64), ("leaf_tests", C.c_uint64), ("ms", C.c_double)]


lib.orc_trace_primary.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Counters)]
lib.orc_trace_rays.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Counters)]
lib.orc_set_cubemap.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
lib.orc_cubemap_color.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
lib.orc_occluded.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
lib.orc_render.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(Counters)]
lib.orc_tri_precompute.argtypes = [C.c_void_p] * 5
lib.orc_camera_from_look.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
lib.orc_tonemap_byte.argtypes = [C.c_float]
lib.orc_tonemap_byte.restype = C.c_ubyte
lib.orc_philox4x32_10.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
lib.orc_sincos.argtypes = [C.c_float, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
lib.orc_hemisphere.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_void_p]
lib.orc_box_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float]
lib.orc_raycheck_rms.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
lib.orc_raycheck_rms.restype = C.c_double


def _soa_ptr(scene):
    return C.cast(scene.soa, C.c_void_p)


def trace_primary(scene, width, height, samples, mode=BRUTE):
    n = width * height * samples * samples
    ids, t, nrm = np.empty(n, np.int32), np.empty(n, np.float32), np.empty((n, 3), np.float32)
    cnt = Counters()
    rc = lib.orc_trace_primary(_soa_ptr(scene), width, height, samples, mode, ids.ctypes.data, t.ctypes.data, nrm.ctypes.data, C.byref(cnt))
    assert rc == 0
    return ids, t, nrm, cnt


def trace_rays(scene, rays, mode=BRUTE):
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
    n = len(rays)
    ids, t, nrm = np.empty(n, np.int32), np.empty(n, np.float32), np.empty((n, 3), np.float32)
    cnt = Counters()
    rc = lib.orc_trace_rays(_soa_ptr(scene), n, rays.ctypes.data, mode, ids.ctypes.data, t.ctypes.data, nrm.ctypes.data, C.byref(cnt))
    assert rc == 0
    return ids, t, nrm, cnt


def set_cubemap(faces):
    """faces: six uint8 arrays [h, w, 3] (+x,-x,+y,-y,+z,-z), row 0 = bottom row; None clears."""
    if faces is None:
        assert lib.orc_set_cubemap(None, None, None) == 0
        return
    faces = [np.ascontiguousarray(f, np.uint8) for f in faces]
    ptrs = (C.c_void_p * 6)(*[f.ctypes.data for f in faces])
    w = (C.c_int * 6)(*[f.shape[1] for f in faces])
    h = (C.c_int * 6)(*[f.shape[0] for f in faces])
    assert lib.orc_set_cubemap(ptrs, w, h) == 0


def cubemap_color(d):
    d = (C.c_float * 3)(*[float(x) for x in d])
    rgba, face, uv = (C.c_float * 4)(), C.c_int(), (C.c_float * 2)()
    assert lib.orc_cubemap_color(d, rgba, C.byref(face), uv) == 0
    return np.array(rgba[:], np.float32), face.value, np.array(uv[:], np.float32)


def occluded(scene, rays, tmax):
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
    tmax = np.ascontiguousarray(tmax, np.float32).reshape(-1)
    out = np.empty(len(rays), np.uint8)
    rc = lib.orc_occluded(_soa_ptr(scene), len(rays), rays.ctypes.data, tmax.ctypes.data, out.ctypes.data)
    assert rc == 0
    return out


def render(scene, params, sincos=SINCOS_POLY, mode=BVH_CANONICAL, threads=0, rows=None):
    """Returns (mean float32[h,w,4], rgb8 uint8[h,w,3], Counters); row 0 = bottom row."""
    w, h = params.width, params.height
    mean, rgb = np.zeros((h, w, 4), np.float32), np.zeros((h, w, 3), np.uint8)
    cnt = Counters()
    y0, y1 = rows if rows else (0, h)
    rc = lib.orc_render(_soa_ptr(scene), C.byref(params), sincos, mode, threads, y0, y1, mean.ctypes.data, rgb.ctypes.data, C.byref(cnt))
    assert rc == 0
    return mean, rgb, cnt


def tri_precompute(verts34):
    v = np.ascontiguousarray(verts34, np.float32).reshape(3, 4)
    out = [np.zeros(4, np.float32) for _ in range(4)]
    ok = lib.orc_tri_precompute(v.ctypes.data, *[o.ctypes.data for o in out])
    return bool(ok), out  # normal, uinv, vinv, ninv


def camera_from_look(viewdir, updir, fov, aspect):
    vd, ud = np.asarray(viewdir, np.float32), np.asarray(updir, np.float32)
    u, v, look = (np.zeros(4, np.float32) for _ in range(3))
    lib.orc_camera_from_look(vd.ctypes.data, ud.ctypes.data, fov, aspect, u.ctypes.data, v.ctypes.data, look.ctypes.data)
    return u, v, look


def tonemap_byte(c):
    return int(lib.orc_tonemap_byte(float(c)))


def philox(ctr, key):
    c, k, o = np.asarray(ctr, np.uint32), np.asarray(key, np.uint32), np.zeros(4, np.uint32)
    lib.orc_philox4x32_10(c.ctypes.data, k.ctypes.data, o.ctypes.data)
    return o


def sincos(p, mode):
    s, c = C.c_float(), C.c_float()
    lib.orc_sincos(float(p), mode, C.byref(s), C.byref(c))
    return s.value, c.value


def hemisphere(normal, u1, u2, mode):
    n, o = np.asarray(normal, np.float32), np.zeros(4, np.float32)
    lib.orc_hemisphere(n.ctypes.data, u1, u2, mode, o.ctypes.data)
    return o


def box_intersect(bmin, bmax, ray6, tmax):
    a, b, r = (np.asarray(x, np.float32) for x in (bmin, bmax, ray6))
    return bool(lib.orc_box_intersect(a.ctypes.data, b.ctypes.data, r.ctypes.data, tmax))


def raycheck_rms(a, b):
    a, b = np.ascontiguousarray(a, np.uint8), np.ascontiguousarray(b, np.uint8)
    assert a.shape == b.shape
    return lib.orc_raycheck_rms(a.ctypes.data, b.ctypes.data, a.size)
