"""pathtracer_amd — MI355X-native backend for the EmeraldShift/pathtracer hot path.

Python here is only a binding layer over the two native libraries built by ``make``
(``lib/libmi_tracer.so``: HIP kernels + C-ABI; ``lib/libpt_host.so``: scene loading,
BVH build, the Tracer interface).  Importing fails if they are not built."""
from ._native import MiError, RenderParams, SceneSoA, Stats  # noqa: F401
from .scene import Scene  # noqa: F401
from .tracer import Context, Group, HipTracer, PinnedFrame, match_cubemap_files, read_image, shard_pixels  # noqa: F401
