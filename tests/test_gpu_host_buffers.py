"""mi_render into the caller's host buffers (what HipTracer::traceImage calls; replaces GpuTracer.cu:181-188): a frame
buffer in pinned host memory (mi_host_alloc, where pt::HipTracer keeps its `buffer`) and a pageable one receive the
same bytes as the oracle's frame."""
import ctypes as C

import numpy as np
import pytest

import pathtracer_amd as P
from pathtracer_amd import _native as N
from oracle import oracle_py as O

pytestmark = pytest.mark.gpu


def test_pinned_and_pageable_frame_buffers_get_the_same_frame(ctx, scenes_small):
    sc = scenes_small["cornell"]
    ctx.upload(sc)
    w, h = sc.image_size(300)
    p = P.Context.params(w, h, 2, 5, seed=8)
    om, orgb, cnt = O.render(sc, p)
    pin = P.PinnedFrame(h, w)
    fb = pin.array
    for _ in range(2):
        fb[:] = 9
        mean, rgb, st = ctx.render(p, rgb_out=fb)
        assert rgb is fb and np.array_equal(fb, orgb) and st.rays_cast == cnt.rays
    assert np.array_equal(mean.view(np.uint32), om.view(np.uint32))        # the pageable float output beside the pinned bytes
    pg = np.full((h, w, 3), 9, np.uint8)
    ctx.render(p, want_mean=False, rgb_out=pg)                             # an ordinary buffer
    assert np.array_equal(pg, orgb)
    # a shard into the pinned buffer: only its own tiles are written
    fb[:] = 0
    for r in range(3):
        ctx.render(P.Context.params(w, h, 2, 5, seed=8, shard=(r, 3)), want_mean=False, rgb_out=fb)
    assert np.array_equal(fb, orgb)
    pin.close()
    out = C.c_void_p()
    assert N.mi.mi_host_alloc(0, C.byref(out)) == N.MI_ERR_INVALID and N.mi.mi_host_alloc(16, None) == N.MI_ERR_INVALID
    N.mi.mi_host_free(None)


def test_tracer_pins_its_frame_buffer_across_setups(scenes_small):
    from pathtracer_amd import scenes
    t = P.HipTracer(depth=2, samples=2, seed=5)
    assert t.load_scene(text=scenes.simple_ray())
    sc = scenes_small["simple"]
    for width in (80, 200, 64):                                            # traceSetup takes new pinned memory when the size changes
        w, h = sc.image_size(width)
        t.trace_setup(w, h)
        for _ in range(2):
            t.trace_image(w, h)
            t.wait_render()
        _, orgb, _ = O.render(sc, P.Context.params(w, h, 2, 2, seed=5))
        assert np.array_equal(t.get_buffer(), orgb)
