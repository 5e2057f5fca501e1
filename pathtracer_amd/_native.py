"""ctypes bindings of the two in-tree native libraries.

* ``libmi_tracer.so`` — the HIP kernels behind the C-ABI of ``include/mi_tracer.h``.
* ``libpt_host.so``   — host side (``.ray`` parser, BVH build, flattening, HipTracer).

There is no Python or CPU fallback: if a library is missing the import fails, and if
no HIP device is present every compute entry point returns an error status that is
raised as :class:`MiError`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MI_TRACER_LIB_DIR: another build of the two libraries (e.g. lib8/: `make wide8`, the 8-wide node format) — never a fallback
LIB_DIR = os.environ.get("MI_TRACER_LIB_DIR") or os.path.join(_HERE, "lib")


class MiError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"mi_tracer status {status}: {message}")
        self.status = status


MI_OK, MI_ERR_INVALID, MI_ERR_NO_DEVICE, MI_ERR_HIP, MI_ERR_NO_SCENE, MI_ERR_OOM = range(6)
MI_RENDER_ONE_STREAM = 1
MI_RENDER_LAUNCH_TIMES = 2
MI_RENDER_ENQUEUE_ONLY = 4

c_float_p = C.POINTER(C.c_float)
c_u32_p = C.POINTER(C.c_uint32)
c_i32_p = C.POINTER(C.c_int32)
c_u8_p = C.POINTER(C.c_uint8)


class SceneSoA(C.Structure):
    """``mi_scene_soa`` (include/mi_tracer.h)."""
    _fields_ = [
        ("n_nodes", C.c_uint32), ("nodes", c_float_p), ("root_ref", C.c_int32),
        ("n_tri", C.c_uint32), ("tri_v0", c_float_p), ("tri_uinv", c_float_p), ("tri_vinv", c_float_p),
        ("tri_ninv", c_float_p), ("tri_normal", c_float_p), ("tri_mat", c_u32_p), ("tri_has_vmat", c_u8_p),
        ("n_sph", C.c_uint32), ("sph_center_radius", c_float_p), ("sph_mat", c_u32_p),
        ("n_mat", C.c_uint32), ("materials", c_float_p),
        ("prim_id", c_i32_p),
        ("cam_eye", C.c_float * 4), ("cam_u", C.c_float * 4), ("cam_v", C.c_float * 4), ("cam_look", C.c_float * 4),
    ]


class RenderParams(C.Structure):
    """``mi_render_params``."""
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("samples_per_dim", C.c_int32), ("depth", C.c_int32),
        ("seed", C.c_uint64), ("tile_w", C.c_int32), ("tile_h", C.c_int32),
        ("shard_rank", C.c_int32), ("shard_count", C.c_int32), ("batch_paths", C.c_int64),
        ("count_work", C.c_int32), ("flags", C.c_int32),
    ]


class Stats(C.Structure):
    """``mi_stats``."""
    _fields_ = [
        ("rays_cast", C.c_uint64), ("paths", C.c_uint64), ("ms_total", C.c_double), ("ms_extend", C.c_double),
        ("ms_shade", C.c_double), ("ms_other", C.c_double), ("extend_launches", C.c_uint32), ("batches", C.c_uint32),
        ("node_visits", C.c_uint64), ("prim_tests", C.c_uint64), ("rays_extend", C.c_uint64), ("ms_tail", C.c_double),
    ]


class CubeMap(C.Structure):
    """``mi_cubemap``."""
    _fields_ = [("width", C.c_int32 * 6), ("height", C.c_int32 * 6), ("rgb", C.c_void_p * 6)]


class BuildStats(C.Structure):
    """``mi_build_stats``."""
    _fields_ = [("ms_build", C.c_double), ("sah_cost", C.c_double), ("depth", C.c_int32), ("n_nodes", C.c_uint32)]


MI_SYMBOLS = [
    "mi_ctx_create", "mi_ctx_destroy", "mi_scene_upload", "mi_scene_upload_build", "mi_scene_quality",
    "mi_scene_download_nodes", "mi_scene_download_nodes4", "mi_scene_download_prim_ids", "mi_cubemap_upload", "mi_render", "mi_render_device", "mi_shard_pixels",
    "mi_assemble_rgb8", "mi_assemble_rgb8_device", "mi_trace_primary", "mi_trace_rays", "mi_occluded_rays", "mi_last_error", "mi_abi_version",
    "mi_render_device_noalloc", "mi_noalloc_workspace_bytes", "mi_host_alloc", "mi_host_free",
    "mi_group_create", "mi_group_destroy", "mi_group_size", "mi_group_scene_upload", "mi_group_scene_upload_build",
    "mi_group_cubemap_upload", "mi_group_render",
]


def _load(name):
    path = os.path.join(LIB_DIR, name)
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: build it with `make` (or __graft_entry__.build()) — there is no fallback")
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


mi = _load("libmi_tracer.so")
host = _load("libpt_host.so")

mi.mi_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
mi.mi_ctx_create.restype = C.c_int
mi.mi_ctx_destroy.argtypes = [C.c_void_p]
mi.mi_ctx_destroy.restype = None
mi.mi_scene_upload.argtypes = [C.c_void_p, C.POINTER(SceneSoA)]
mi.mi_scene_upload.restype = C.c_int
mi.mi_render.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_void_p, C.c_void_p, C.POINTER(Stats)]
mi.mi_render.restype = C.c_int
mi.mi_render_device.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Stats)]
mi.mi_render_device.restype = C.c_int
mi.mi_render_device_noalloc.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Stats)]
mi.mi_render_device_noalloc.restype = C.c_int
mi.mi_noalloc_workspace_bytes.restype = C.c_int64
mi.mi_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
mi.mi_host_alloc.restype = C.c_int
mi.mi_host_free.argtypes = [C.c_void_p]
mi.mi_host_free.restype = None
mi.mi_shard_pixels.argtypes = [C.POINTER(RenderParams)]
mi.mi_shard_pixels.restype = C.c_int64
mi.mi_assemble_rgb8.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_void_p, C.c_int64, C.c_void_p]
mi.mi_assemble_rgb8.restype = C.c_int
mi.mi_trace_primary.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
mi.mi_trace_primary.restype = C.c_int
mi.mi_assemble_rgb8_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
mi.mi_assemble_rgb8_device.restype = C.c_int
mi.mi_trace_rays.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
mi.mi_trace_rays.restype = C.c_int
mi.mi_occluded_rays.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
mi.mi_occluded_rays.restype = C.c_int
mi.mi_scene_upload_build.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(BuildStats)]
mi.mi_scene_upload_build.restype = C.c_int
mi.mi_scene_download_nodes4.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_int32),
                                        C.POINTER(C.c_int32), C.POINTER(C.c_float)]
mi.mi_scene_download_nodes4.restype = C.c_int
mi.mi_scene_download_prim_ids.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
mi.mi_scene_download_prim_ids.restype = C.c_int
mi.mi_cubemap_upload.argtypes = [C.c_void_p, C.c_void_p]
mi.mi_cubemap_upload.restype = C.c_int
mi.mi_scene_quality.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
mi.mi_scene_quality.restype = C.c_int
mi.mi_scene_download_nodes.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_int32)]
mi.mi_scene_download_nodes.restype = C.c_int
mi.mi_group_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
mi.mi_group_create.restype = C.c_int
mi.mi_group_destroy.argtypes = [C.c_void_p]
mi.mi_group_destroy.restype = None
mi.mi_group_size.argtypes = [C.c_void_p]
mi.mi_group_size.restype = C.c_int
mi.mi_group_scene_upload.argtypes = [C.c_void_p, C.POINTER(SceneSoA)]
mi.mi_group_scene_upload.restype = C.c_int
mi.mi_group_scene_upload_build.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(BuildStats)]
mi.mi_group_scene_upload_build.restype = C.c_int
mi.mi_group_cubemap_upload.argtypes = [C.c_void_p, C.c_void_p]
mi.mi_group_cubemap_upload.restype = C.c_int
mi.mi_group_render.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_void_p, C.POINTER(Stats)]
mi.mi_group_render.restype = C.c_int
mi.mi_last_error.restype = C.c_char_p
mi.mi_abi_version.restype = C.c_int

host.pt_last_error.restype = C.c_char_p
host.pt_scene_load.argtypes = [C.c_char_p, C.c_int]
host.pt_scene_load.restype = C.c_void_p
host.pt_scene_free.argtypes = [C.c_void_p]
host.pt_scene_soa.argtypes = [C.c_void_p]
host.pt_scene_soa.restype = C.POINTER(SceneSoA)
host.pt_scene_slot_vertices.argtypes = [C.c_void_p]
host.pt_scene_slot_vertices.restype = c_float_p
host.pt_scene_aspect.argtypes = [C.c_void_p]
host.pt_scene_aspect.restype = C.c_double
host.pt_scene_bvh_depth.argtypes = [C.c_void_p]
host.pt_scene_bvh_depth.restype = C.c_int
host.pt_scene_sah_cost.argtypes = [C.c_void_p]
host.pt_scene_sah_cost.restype = C.c_double
host.pt_scene_new.restype = C.c_void_p
host.pt_scene_add_material.argtypes = [C.c_void_p, c_float_p]
host.pt_scene_add_material.restype = C.c_uint32
host.pt_scene_add_mesh.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_uint32]
host.pt_scene_add_mesh.restype = C.c_int64
host.pt_scene_add_sphere.argtypes = [C.c_void_p, c_float_p, C.c_float, C.c_uint32]
host.pt_scene_set_camera.argtypes = [C.c_void_p, c_float_p, c_float_p, c_float_p, C.c_float, C.c_float]
host.pt_scene_finalize.argtypes = [C.c_void_p]
host.pt_hip_tracer_new.argtypes = [C.c_int, C.c_int, C.c_int, C.c_ulonglong]
host.pt_hip_tracer_new_group.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_ulonglong]
host.pt_hip_tracer_new_group.restype = C.c_void_p
host.pt_hip_tracer_new.restype = C.c_void_p
host.pt_tracer_free.argtypes = [C.c_void_p]
host.pt_tracer_load_scene.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
host.pt_tracer_load_scene.restype = C.c_int
host.pt_tracer_load_cubemap.argtypes = [C.c_void_p, C.c_char_p]
host.pt_tracer_load_cubemap.restype = C.c_int
host.pt_read_image.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_size_t]
host.pt_read_image.restype = C.c_int
host.pt_match_cubemap.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
host.pt_match_cubemap.restype = C.c_int
host.pt_tracer_aspect.argtypes = [C.c_void_p]
host.pt_tracer_aspect.restype = C.c_double
host.pt_tracer_trace_setup.argtypes = [C.c_void_p, C.c_int, C.c_int]
host.pt_tracer_trace_image.argtypes = [C.c_void_p, C.c_int, C.c_int]
host.pt_tracer_wait_render.argtypes = [C.c_void_p]
host.pt_tracer_wait_render.restype = C.c_int
host.pt_tracer_get_buffer.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_ubyte)), C.POINTER(C.c_int), C.POINTER(C.c_int)]
host.pt_tracer_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
host.pt_write_image.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p]
host.pt_write_image.restype = C.c_int


def check(status):
    if status != MI_OK:
        raise MiError(status, mi.mi_last_error().decode())
