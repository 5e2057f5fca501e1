"""The per-level secondary kernels under test at every size.

By default k_tail takes over a batch once a level holds at most 200-700 k rays (1/40 of the batch's paths;
MI_TRACER_TAIL sets a fixed number), so in every small test the levels >= 1 never reach k_extend<false,false> /
k_shade<false>.  This module repeats the parity,
golden and fuzz comparisons in contexts of their own (the tuning environment is read in mi_ctx_create):
  MI_TRACER_TAIL=0     every level through the per-level kernels (sub-streams, segcap arithmetic,
                       k_shade<false> compaction), k_tail never runs;
  MI_TRACER_TAIL=2000  the hand-over to k_tail happens in the middle of the paths;
  MI_TRACER_SPLIT_MIN=1000  frames of more than 2000 paths are cut into two batches that run on the context's two
                       streams (by default only frames of several million paths are), MI_TRACER_STREAMS=1: one stream;
and runs k_tail on a grid LARGER than k_extend's (MI_TRACER_TAIL_WG_PER_CU=16) on a scene whose
traversal stacks spill into the global overflow stack.  Reference path: CpuTracer.cu:99-186."""
import ctypes as C
import os

import numpy as np
import pytest

import pathtracer_amd as P
from pathtracer_amd import _native as N
from pathtracer_amd import scenes
from oracle import oracle_py as O
from gpu_util import fresh_context, fuzz_configurations, same

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module", params=[dict(MI_TRACER_TAIL=0), dict(MI_TRACER_TAIL=2000), dict(MI_TRACER_TAIL=2000, MI_TRACER_SPLIT_MIN=1000),
                                        dict(MI_TRACER_STREAMS=1)],
                ids=["tail_off", "tail_2000", "two_streams_at_every_size", "one_stream"])
def ctx_tail(request):
    with fresh_context(**request.param) as c:
        yield c


@pytest.mark.parametrize("name,width,samples,depth", [("simple", 64, 2, 2), ("cornell", 48, 2, 5), ("glass", 64, 1, 8), ("blob3", 48, 2, 3)])
def test_image_byte_exact_vs_oracle(ctx_tail, scenes_small, name, width, samples, depth):
    sc = scenes_small[name]
    w, h = sc.image_size(width)
    ctx_tail.upload(sc)
    p = P.Context.params(w, h, samples, depth, seed=7)
    mean, rgb, st = ctx_tail.render(p)
    omean, orgb, cnt = O.render(sc, p)
    assert st.rays_cast == cnt.rays and st.paths == w * h * samples * samples
    assert same(mean, omean) and np.array_equal(rgb, orgb)


@pytest.mark.parametrize("name", ["simple", "cornell", "glass", "blob2"])
def test_hip_reproduces_golden(ctx_tail, name):
    g = np.load(os.path.join(GOLD, f"{name}.npz"))
    sc = P.Scene.from_text(str(g["text"]))
    w, h, samples, depth, seed = [int(x) for x in g["size"]]
    ctx_tail.upload(sc)
    mean, rgb, st = ctx_tail.render(P.Context.params(w, h, samples, depth, seed=seed))
    assert st.rays_cast == int(g["rays"][0])
    assert np.array_equal(rgb, g["rgb"]) and same(mean, g["mean"])


@pytest.mark.parametrize("seed", [21, 22])
def test_random_configurations_match_the_oracle(ctx_tail, seed):
    fuzz_configurations(ctx_tail, seed, 30)


def test_mid_size_frames_every_level_in_the_wavefront_kernels(ctx_tail):
    """Whole frames large enough that every sub-stream and several persistent waves per CU carry rays at
    levels >= 1 (0.6 M and 1 M paths), compared with the oracle in full."""
    for sc, width, samples, depth in ((P.Scene.from_text(scenes.blob_ray(4)), 400, 2, 5),
                                      (P.Scene.from_text(scenes.glass_ray()), 360, 2, 8),
                                      (P.Scene.from_text(scenes.cornell_ray()), 256, 4, 5)):
        w, h = sc.image_size(width)
        ctx_tail.upload(sc)
        p = P.Context.params(w, h, samples, depth, seed=5)
        mean, rgb, st = ctx_tail.render(p)
        omean, orgb, cnt = O.render(sc, p)
        assert st.rays_cast == cnt.rays
        assert same(mean, omean) and np.array_equal(rgb, orgb)
        # the same frame in three batches: stream buffers are reused and re-counted per batch
        mean2, rgb2, st2 = ctx_tail.render(P.Context.params(w, h, samples, depth, seed=5, batch_paths=w * h * samples * samples // 3 + 1))
        assert st2.batches >= 3 and st2.rays_cast == cnt.rays and same(mean2, omean)


def _glass_chain(levels):
    """The chain scene of test_gpu_edge_cases (a BVH that forces a traversal stack of ~1.5 x `levels` entries for
    rays running down -z) with a material every ray passes straight through (kt = 1, index = 1): each path
    crosses one pair of triangles per bounce, so the SECONDARY rays start high up and walk the deep chain."""
    from test_gpu_edge_cases import _chain_scene
    s, keep = _chain_scene(levels)
    keep["mats"][:] = np.array(scenes.mat20(diffuse=(0.3, 0.3, 0.3), transmissive=(1, 1, 1), index=1.0), np.float32)
    s.cam_u[:] = [0.2, 0, 0, 0]     # a narrow view: most camera rays enter the column of triangles and stay in it
    s.cam_v[:] = [0, 0.2, 0, 0]
    return s, keep


def test_tail_grid_larger_than_the_extend_grid_with_deep_stacks():
    """the overflow stacks were sized for k_extend's grid while k_tail indexes them with its own,
    which MI_TRACER_TAIL_WG_PER_CU can make larger.  0.6 M paths: the static chunk deal reaches workgroups
    beyond k_extend's grid; 62 chain levels: ~90 stack entries per ray against 12 in LDS."""
    s, keep = _glass_chain(62)

    class Fake:  # the oracle binding only needs .soa
        soa = C.pointer(s)
    with fresh_context(MI_TRACER_TAIL_WG_PER_CU=16, MI_TRACER_TAIL=700000) as c:  # an explicit threshold: every level >= 1 of this frame in k_tail
        N.check(N.mi.mi_scene_upload(c._h, C.byref(s)))
        p = P.Context.params(768, 768, 1, 8, seed=1)
        mean, rgb, st = c.render(p)
        omean, orgb, cnt = O.render(Fake, p)
        assert st.rays_cast == cnt.rays and st.rays_cast > 3 * st.paths     # the paths really go deep
        assert st.rays_extend == st.paths                                   # level >= 1 ran in k_tail
        assert same(mean, omean) and np.array_equal(rgb, orgb)


def test_small_memory_budget_cuts_frames_into_equal_batches(ctx, scenes_small):
    # MI_TRACER_BATCH_GB=1: 3.3 M path slots per batch at depth 5.  Frames of 2, 3 (made 4 for the two streams) and 5
    # default-sized batches must equal the one-batch render of the default context, ray for ray.
    sc = scenes_small["blob3"]
    ctx.upload(sc)
    with fresh_context(MI_TRACER_BATCH_GB=1) as small:
        small.upload(sc)
        for width, s, want_batches in ((1600, 2, 2), (1920, 2, 4), (1600, 3, 4), (1920, 3, 6)):
            w, h = sc.image_size(width)
            p = P.Context.params(w, h, s, 5, seed=2)
            mean, rgb, st = ctx.render(p)
            mean_s, rgb_s, st_s = small.render(p)
            assert st_s.batches == want_batches and st.batches <= 2, (width, s, st_s.batches, st.batches)
            assert st_s.rays_cast == st.rays_cast and st_s.paths == st.paths
            assert np.array_equal(rgb_s, rgb) and same(mean_s, mean)
