"""Python mirror of the reference's Tracer interface for the MI355X backend.

``HipTracer`` has the reference's method sequence (src/tracer.h:19-52, driven by
src/ui/CommandLineUI.cu:82-94): ``load_scene -> trace_setup -> trace_image ->
wait_render -> get_buffer``.  ``Context`` is the thin object over the raw C-ABI
(``include/mi_tracer.h``) used by the parity tests and the bench.
"""
import ctypes as C

import numpy as np

from . import _native as N
from .scene import Scene


class Context:
    """One ``mi_ctx`` = one GPU."""

    def __init__(self, device=0):
        h = C.c_void_p()
        N.check(N.mi.mi_ctx_create(device, C.byref(h)))
        self._h = h
        self.scene = None

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h and N is not None and getattr(N, "mi", None) is not None:  # module globals may be gone at interpreter exit
            N.mi.mi_ctx_destroy(h)

    __del__ = close

    def upload(self, scene: Scene):
        N.check(N.mi.mi_scene_upload(self._h, scene.soa))
        self.scene = scene  # keep the host arrays alive (not required by the ABI)

    def upload_build(self, scene: Scene):
        """Upload the primitives and build the BVH on the device (mi_scene_upload_build)."""
        verts = np.ascontiguousarray(scene.slot_vertices()[:, :, :3], np.float32)
        st = N.BuildStats()
        N.check(N.mi.mi_scene_upload_build(self._h, scene.soa, verts.ctypes.data if len(verts) else None, C.byref(st)))
        self.scene = scene
        return st

    def set_cubemap(self, faces):
        """faces: six uint8 arrays [h, w, 3] (+x,-x,+y,-y,+z,-z), row 0 = bottom row; None removes it."""
        if faces is None:
            N.check(N.mi.mi_cubemap_upload(self._h, None))
            return
        faces = [np.ascontiguousarray(f, np.uint8) for f in faces]
        if len(faces) != 6 or any(f.ndim != 3 or f.shape[2] != 3 for f in faces):
            raise ValueError("six [h, w, 3] uint8 faces")
        cm = N.CubeMap()
        for i, f in enumerate(faces):
            cm.width[i], cm.height[i], cm.rgb[i] = f.shape[1], f.shape[0], f.ctypes.data
        N.check(N.mi.mi_cubemap_upload(self._h, C.byref(cm)))

    def scene_quality(self):
        v = C.c_double()
        N.check(N.mi.mi_scene_quality(self._h, C.byref(v)))
        return v.value

    def download_nodes(self):
        """(float32[n_nodes,16] nodes, root_ref) of the tree resident on the device."""
        n, root = C.c_uint32(), C.c_int32()
        N.check(N.mi.mi_scene_download_nodes(self._h, None, 0, C.byref(n), C.byref(root)))
        nodes = np.zeros((n.value, 16), np.float32)
        N.check(N.mi.mi_scene_download_nodes(self._h, nodes.ctypes.data if n.value else None, n.value, C.byref(n), C.byref(root)))
        return nodes, root.value

    def download_nodes4(self):
        """The tree the kernels walk: (fp32 nodes float32[n,32], quantised nodes uint32[n,16], root_ref,
        treelet_base, qs float32[3], qb float32[3])."""
        n, root, tb = C.c_uint32(0), C.c_int32(0), C.c_int32(0)
        grid = (C.c_float * 6)()
        N.check(N.mi.mi_scene_download_nodes4(self._h, None, None, 0, C.byref(n), C.byref(root), C.byref(tb), grid))
        f = np.zeros((n.value, 32), np.float32)
        q = np.zeros((n.value, 16), np.uint32)
        if n.value:
            N.check(N.mi.mi_scene_download_nodes4(self._h, f.ctypes.data, q.ctypes.data, n.value, None, None, None, None))
        g = np.array(list(grid), np.float32)
        return f, q, root.value, tb.value, g[:3], g[3:]

    def download_prim_ids(self):
        """int32[n_prims]: primitive id of every device slot (permuted by upload_build)."""
        n = self.scene.n_prims
        ids = np.zeros(n, np.int32)
        if n:
            N.check(N.mi.mi_scene_download_prim_ids(self._h, ids.ctypes.data, n))
        return ids

    @staticmethod
    def params(width, height, samples, depth, seed=0, tile=(0, 0), shard=(0, 1), batch_paths=0, count_work=False, one_stream=False, launch_times=False,
               enqueue_only=False):
        return N.RenderParams(width, height, samples, depth, seed, tile[0], tile[1], shard[0], shard[1], batch_paths,
                              1 if count_work else 0, (N.MI_RENDER_ONE_STREAM if one_stream else 0) | (N.MI_RENDER_LAUNCH_TIMES if launch_times else 0)
                              | (N.MI_RENDER_ENQUEUE_ONLY if enqueue_only else 0))

    def render(self, params, want_mean=True, want_rgb8=True, rgb_out=None):
        """Host-buffer render.  Returns (mean float32[h,w,4] | None, rgb8 uint8[h,w,3] | None, Stats);
        row 0 is the bottom row (src/tracer.cu:96).  rgb_out: a caller-owned uint8[h,w,3] array to render into
        (e.g. one pinned with host_register, as pt::HipTracer pins its frame buffer)."""
        w, h = params.width, params.height
        mean = np.zeros((h, w, 4), np.float32) if want_mean else None
        if rgb_out is not None:
            if rgb_out.dtype != np.uint8 or rgb_out.shape != (h, w, 3) or not rgb_out.flags.c_contiguous:
                raise ValueError("rgb_out must be a contiguous uint8[h, w, 3] array")
            want_rgb8 = True
        rgb = rgb_out if rgb_out is not None else (np.zeros((h, w, 3), np.uint8) if want_rgb8 else None)
        st = N.Stats()
        N.check(N.mi.mi_render(self._h, C.byref(params), mean.ctypes.data if want_mean else None,
                               rgb.ctypes.data if want_rgb8 else None, C.byref(st)))
        return mean, rgb, st

    def render_device(self, params, d_mean_ptr, d_rgba8_ptr, stream=None):
        """mi_render_device.  With params built with enqueue_only=True the frame is only enqueued on `stream` (the
        caller synchronises) and None is returned instead of the Stats."""
        if params.flags & N.MI_RENDER_ENQUEUE_ONLY:
            N.check(N.mi.mi_render_device(self._h, C.byref(params), d_mean_ptr, d_rgba8_ptr, stream, None))
            return None
        st = N.Stats()
        N.check(N.mi.mi_render_device(self._h, C.byref(params), d_mean_ptr, d_rgba8_ptr, stream, C.byref(st)))
        return st

    def render_device_noalloc(self, params, d_mean_ptr, d_rgba8_ptr, stream=None, want_stats=True):
        """mi_render_device_noalloc: same outputs from the static workspace, no allocation.  want_stats=False only
        enqueues the work on `stream` (the caller synchronises)."""
        st = N.Stats() if want_stats else None
        N.check(N.mi.mi_render_device_noalloc(self._h, C.byref(params), d_mean_ptr, d_rgba8_ptr, stream, C.byref(st) if want_stats else None))
        return st

    def assemble_rgb8(self, params, d_gathered_ptr, shard_stride_pixels):
        rgb = np.zeros((params.height, params.width, 3), np.uint8)
        N.check(N.mi.mi_assemble_rgb8(self._h, C.byref(params), d_gathered_ptr, shard_stride_pixels, rgb.ctypes.data))
        return rgb

    def assemble_rgb8_device(self, params, d_gathered_ptr, shard_stride_pixels, d_rgb8_ptr, stream=None):
        N.check(N.mi.mi_assemble_rgb8_device(self._h, C.byref(params), d_gathered_ptr, shard_stride_pixels, d_rgb8_ptr, stream))

    def trace_primary(self, width, height, samples, want_normals=True):
        n = width * height * samples * samples
        ids = np.empty(n, np.int32)
        t = np.empty(n, np.float32)
        nrm = np.empty((n, 3), np.float32) if want_normals else None
        N.check(N.mi.mi_trace_primary(self._h, width, height, samples, ids.ctypes.data, t.ctypes.data,
                                      nrm.ctypes.data if want_normals else None))
        return ids, t, nrm

    def trace_rays(self, rays, want_normals=True):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        n = len(rays)
        ids = np.empty(n, np.int32)
        t = np.empty(n, np.float32)
        nrm = np.empty((n, 3), np.float32) if want_normals else None
        N.check(N.mi.mi_trace_rays(self._h, n, rays.ctypes.data, ids.ctypes.data, t.ctypes.data,
                                   nrm.ctypes.data if want_normals else None))
        return ids, t, nrm

    def occluded_rays(self, rays, tmax):
        """Any-hit query (mi_occluded_rays): uint8[n], 1 = something is hit within tmax."""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        tmax = np.ascontiguousarray(tmax, np.float32).reshape(-1)
        if len(tmax) != len(rays):
            raise ValueError("one tmax per ray")
        out = np.empty(len(rays), np.uint8)
        N.check(N.mi.mi_occluded_rays(self._h, len(rays), rays.ctypes.data, tmax.ctypes.data, out.ctypes.data))
        return out


class Group:
    """One ``mi_group``: several GPUs in one process (scene replicated, frame sharded by tile, RGBA8 tiles gathered to
    the first device over RCCL)."""

    def __init__(self, devices=(0,)):
        ids = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        N.check(N.mi.mi_group_create(ids, len(devices), C.byref(h)))
        self._h = h
        self.scene = None

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h and N is not None and getattr(N, "mi", None) is not None:
            N.mi.mi_group_destroy(h)

    __del__ = close

    def __len__(self):
        return N.mi.mi_group_size(self._h)

    def upload(self, scene):
        N.check(N.mi.mi_group_scene_upload(self._h, scene.soa))
        self.scene = scene

    def upload_build(self, scene):
        verts = np.ascontiguousarray(scene.slot_vertices()[:, :, :3], np.float32)
        st = N.BuildStats()
        N.check(N.mi.mi_group_scene_upload_build(self._h, scene.soa, verts.ctypes.data if len(verts) else None, C.byref(st)))
        self.scene = scene
        return st

    def set_cubemap(self, faces):
        if faces is None:
            N.check(N.mi.mi_group_cubemap_upload(self._h, None))
            return
        faces = [np.ascontiguousarray(f, np.uint8) for f in faces]
        cm = N.CubeMap()
        for i, f in enumerate(faces):
            cm.width[i], cm.height[i], cm.rgb[i] = f.shape[1], f.shape[0], f.ctypes.data
        N.check(N.mi.mi_group_cubemap_upload(self._h, C.byref(cm)))

    def render(self, params):
        """(rgb8 uint8[h,w,3], Stats); row 0 is the bottom row."""
        rgb = np.zeros((params.height, params.width, 3), np.uint8)
        st = N.Stats()
        N.check(N.mi.mi_group_render(self._h, C.byref(params), rgb.ctypes.data, C.byref(st)))
        return rgb, st


class PinnedFrame:
    """uint8[h, w, 3] frame buffer in pinned host memory (mi_host_alloc), as pt::HipTracer keeps its `buffer`:
    ``.array`` is a numpy view; ``close()`` releases the memory (the view must not be used afterwards)."""

    def __init__(self, h, w):
        p = C.c_void_p()
        N.check(N.mi.mi_host_alloc(h * w * 3, C.byref(p)))
        self._p = p
        self.array = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_ubyte)), shape=(h, w, 3))

    def close(self):
        p, self._p = getattr(self, "_p", None), None
        self.array = None
        if p and N is not None and getattr(N, "mi", None) is not None:
            N.mi.mi_host_free(p)

    __del__ = close


def shard_pixels(params):
    return N.mi.mi_shard_pixels(C.byref(params))


class HipTracer:
    """The Tracer-shaped object (C++ class pt::HipTracer in pathtracer_amd/host/tracer.h)."""

    def __init__(self, depth=0, samples=2, device=0, seed=0, n_devices=0):
        """n_devices > 0: the devices device .. device + n_devices - 1 as one mi_group (`ray -N`)."""
        h = (N.host.pt_hip_tracer_new_group(depth, samples, device, n_devices, seed) if n_devices > 0
             else N.host.pt_hip_tracer_new(depth, samples, device, seed))
        if not h:
            raise N.MiError(N.MI_ERR_NO_DEVICE, N.host.pt_last_error().decode())
        self._h = C.c_void_p(h)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            N.host.pt_tracer_free(h)

    def load_scene(self, path=None, text=None):
        arg = (str(path).encode(), 1) if path is not None else (text.encode(), 0)
        return bool(N.host.pt_tracer_load_scene(self._h, *arg))

    def load_cubemap(self, one_face_file):
        """TraceUI::smartLoadCubemap: one of the six face files names the directory."""
        return bool(N.host.pt_tracer_load_cubemap(self._h, str(one_face_file).encode()))

    def aspect_ratio(self):
        return N.host.pt_tracer_aspect(self._h)

    def trace_setup(self, w, h):
        N.host.pt_tracer_trace_setup(self._h, w, h)

    def trace_image(self, w, h):
        N.host.pt_tracer_trace_image(self._h, w, h)

    def wait_render(self):
        if not N.host.pt_tracer_wait_render(self._h):
            raise N.MiError(N.MI_ERR_HIP, N.host.pt_last_error().decode())

    def get_buffer(self):
        buf = C.POINTER(C.c_ubyte)()
        w, h = C.c_int(), C.c_int()
        N.host.pt_tracer_get_buffer(self._h, C.byref(buf), C.byref(w), C.byref(h))
        return np.ctypeslib.as_array(buf, shape=(h.value, w.value, 3)).copy()

    def stats(self):
        st = N.Stats()
        N.host.pt_tracer_stats(self._h, C.byref(st))
        return st


def read_image(path):
    """readImage of the host library: uint8[h, w, 3], row 0 = bottom row; None if unreadable."""
    w, h = C.c_int(), C.c_int()
    if not N.host.pt_read_image(str(path).encode(), C.byref(w), C.byref(h), None, 0):
        return None
    out = np.empty((h.value, w.value, 3), np.uint8)
    assert N.host.pt_read_image(str(path).encode(), C.byref(w), C.byref(h), out.ctypes.data, out.nbytes)
    return out


def match_cubemap_files(one_face_file):
    """matchCubemapFiles of the host library: the six file names (+x,-x,+y,-y,+z,-z) or None."""
    buf = C.create_string_buffer(8192)
    if not N.host.pt_match_cubemap(str(one_face_file).encode(), buf, len(buf)):
        return None
    return buf.value.decode().split("\n")
