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


def test_enqueue_only_frames_match_waited_frames(ctx, scenes_small):
    # MI_RENDER_ENQUEUE_ONLY: frames enqueued back to back on one stream, one wait at the end (bench.py's timed loop); growing
    # batches in between (a larger frame after a smaller one) must not free buffers a running frame still uses
    import ctypes as C
    import torch
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    sc = scenes_small["glass"]
    ctx.upload(sc)
    outs, wants = [], []
    for width, s, d in ((64, 1, 3), (200, 2, 8), (96, 2, 5), (333, 2, 8)):
        w, h = sc.image_size(width)
        p = P.Context.params(w, h, s, d, seed=13)
        n = P.shard_pixels(p)
        want = torch.zeros((n, 4), dtype=torch.uint8, device=dev)
        ctx.render_device(p, None, want.data_ptr(), stream.cuda_stream)
        wants.append(want)
    c2 = P.Context(0)                                                      # a fresh context: its buffers start empty and must grow
    c2.upload(sc)
    for width, s, d in ((64, 1, 3), (200, 2, 8), (96, 2, 5), (333, 2, 8)):
        w, h = sc.image_size(width)
        pq = P.Context.params(w, h, s, d, seed=13, enqueue_only=True)
        out = torch.full((P.shard_pixels(pq), 4), 3, dtype=torch.uint8, device=dev)
        assert c2.render_device(pq, None, out.data_ptr(), stream.cuda_stream) is None
        outs.append(out)
    stream.synchronize()
    for out, want in zip(outs, wants):
        assert torch.equal(out, want)
    # frames large enough for the context's two streams: three in a row into three buffers, one wait; a consumer's stream
    # ordered behind them by an event of the caller's
    sc2 = scenes_small["cornell"]
    c2.upload(sc2)
    w, h = sc2.image_size(512)
    p = P.Context.params(w, h, 6, 5, seed=2)
    n = P.shard_pixels(p)
    want = torch.zeros((n, 4), dtype=torch.uint8, device=dev)
    stw = c2.render_device(p, None, want.data_ptr(), stream.cuda_stream)
    assert stw.batches == 2
    pq = P.Context.params(w, h, 6, 5, seed=2, enqueue_only=True)
    bufs = [torch.full((n, 4), 5, dtype=torch.uint8, device=dev) for _ in range(3)]
    for bf in bufs:
        c2.render_device(pq, None, bf.data_ptr(), stream.cuda_stream)
    consumer = torch.cuda.Stream(device=dev)
    consumer.wait_stream(stream)
    with torch.cuda.stream(consumer):
        copies = [bf.clone() for bf in bufs]                               # ordered behind all three frames by the flush
    consumer.synchronize()
    for cp in copies:
        assert torch.equal(cp, want)
    st2 = c2.render_device(p, None, bufs[0].data_ptr(), stream.cuda_stream)   # a waiting call closes whatever is open
    assert st2.rays_cast == stw.rays_cast and torch.equal(bufs[0], want)
    st = N.Stats()
    pq = P.Context.params(32, 32, 1, 1, enqueue_only=True)
    assert N.mi.mi_render_device(c2._h, C.byref(pq), None, outs[0].data_ptr(), None, C.byref(st)) == N.MI_ERR_INVALID   # no stats without a wait
    pc = P.Context.params(32, 32, 1, 1, enqueue_only=True, count_work=True)
    assert N.mi.mi_render_device(c2._h, C.byref(pc), None, outs[0].data_ptr(), None, None) == N.MI_ERR_INVALID
    rgb = np.zeros((32, 32, 3), np.uint8)
    assert N.mi.mi_render(c2._h, C.byref(pq), None, rgb.ctypes.data, None) == N.MI_ERR_INVALID                          # host outputs need the wait
    c2.close()
