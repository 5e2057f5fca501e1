"""mi_render_device_noalloc — the one entry point that allocates nothing (static 64-MiB workspace) — against
mi_render_device, through the C-ABI, bit for bit: two calls in a row, the second with a smaller input (it finds
the first call's contents in the workspace), single- and multi-batch frames, the enqueue-only mode, a device-built
tree, a shard, and the rejection of inputs whose smallest batch does not fit.  Its defining contract — no allocation,
usable under HIP graph capture — is exercised too: free device memory is unchanged across the calls, and the
enqueue-only mode is captured into a graph (capture fails on any allocation or synchronisation) and replayed twice."""
import ctypes as C

import numpy as np
import pytest
import torch

import pathtracer_amd as P
from pathtracer_amd import _native as N
from pathtracer_amd import scenes

pytestmark = pytest.mark.gpu


def both(ctx, p, stream=None, want_stats=True):
    n = P.shard_pixels(p)
    dev = torch.device("cuda", 0)
    outs = []
    for noalloc in (False, True):
        mean = torch.full((n, 4), -1.0, dtype=torch.float32, device=dev)
        rgba = torch.full((n, 4), 7, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        if noalloc:
            st = ctx.render_device_noalloc(p, mean.data_ptr(), rgba.data_ptr(), stream, want_stats)
        else:
            st = ctx.render_device(p, mean.data_ptr(), rgba.data_ptr(), stream)
        torch.cuda.synchronize()   # the enqueue-only mode returns before the work is done
        outs.append((mean.cpu().numpy(), rgba.cpu().numpy(), st))
    (m0, r0, s0), (m1, r1, s1) = outs
    assert np.array_equal(m0.view(np.uint32), m1.view(np.uint32)), "mean radiance differs"
    assert np.array_equal(r0, r1), "RGBA8 differs"
    if want_stats:
        assert s1.rays_cast == s0.rays_cast and s1.paths == s0.paths
    return s0, s1


def test_noalloc_matches_render_device_twice_in_a_row(ctx, scenes_small):
    assert N.mi.mi_noalloc_workspace_bytes() == 64 << 20
    sc = scenes_small["cornell"]
    ctx.upload(sc)
    w, h = sc.image_size(512)
    s0, s1 = both(ctx, P.Context.params(w, h, 2, 5, seed=3))          # 1 M paths: several batches from the workspace
    assert s1.batches > 1 and s0.batches == 1
    w2, h2 = sc.image_size(96)
    s0, s1 = both(ctx, P.Context.params(w2, h2, 2, 5, seed=3))        # smaller input, one batch, stale workspace contents
    assert s1.batches == 1
    both(ctx, P.Context.params(w2, h2, 1, -1, seed=9))                # primary rays only
    both(ctx, P.Context.params(33, 17, 3, 8, seed=4, tile=(5, 7), shard=(1, 3)))  # ragged tiles, a shard


def test_noalloc_other_scenes_and_enqueue_only(ctx, scenes_small):
    stream = torch.cuda.Stream()
    for name, width, s, d in (("glass", 200, 2, 8), ("blob3", 160, 2, 4), ("simple", 128, 1, 2)):
        sc = scenes_small[name]
        ctx.upload(sc)
        w, h = sc.image_size(width)
        both(ctx, P.Context.params(w, h, s, d, seed=11), stream.cuda_stream, want_stats=False)
    sc = scenes.blob_scene(5)                                          # 20 k triangles, device-built tree (deeper)
    ctx.upload_build(sc)
    w, h = sc.image_size(640)
    s0, s1 = both(ctx, P.Context.params(w, h, 2, 5, seed=2))
    assert s1.batches > 1
    ctx.upload(sc)
    both(ctx, P.Context.params(w, h, 1, 3, seed=2))


def test_noalloc_rejects_what_cannot_fit(ctx, scenes_small):
    ctx.upload(scenes_small["simple"])
    p = P.Context.params(64, 64, 200, 5)                               # 64 pixels x 40000 samples: no batch fits 64 MiB
    out = torch.zeros(P.shard_pixels(p) * 4, dtype=torch.uint8, device="cuda:0")
    rc = N.mi.mi_render_device_noalloc(ctx._h, C.byref(p), None, out.data_ptr(), None, None)
    assert rc == N.MI_ERR_INVALID and b"static workspace" in N.mi.mi_last_error()
    pc = P.Context.params(16, 16, 1, 1, count_work=True)
    assert N.mi.mi_render_device_noalloc(ctx._h, C.byref(pc), None, out.data_ptr(), None, None) == N.MI_ERR_INVALID
    # and the regular entry point still takes it
    ctx.render(P.Context.params(16, 16, 2, 1))


def test_noalloc_allocates_nothing_and_runs_under_graph_capture(ctx, scenes_small):
    sc = scenes_small["cornell"]
    ctx.upload(sc)
    w, h = sc.image_size(160)
    p = P.Context.params(w, h, 2, 5, seed=21)
    n = P.shard_pixels(p)
    dev = torch.device("cuda", 0)
    want_mean = torch.zeros((n, 4), dtype=torch.float32, device=dev)
    want_rgba = torch.zeros((n, 4), dtype=torch.uint8, device=dev)
    ctx.render_device(p, want_mean.data_ptr(), want_rgba.data_ptr())
    mean = torch.zeros((n, 4), dtype=torch.float32, device=dev)
    rgba = torch.zeros((n, 4), dtype=torch.uint8, device=dev)
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):                                       # the runtime backs a new stream with a hardware queue on first use
        rgba.zero_()
    # ... and grows pools of its own (signals, kernel-argument memory, queues that other tests' streams left) when a stream
    # first carries this many launches: one un-measured pass over both modes first, so that the measurement below sees only
    # what the ENTRY POINT allocates
    ctx.render_device_noalloc(p, mean.data_ptr(), rgba.data_ptr(), None, True)
    ctx.render_device_noalloc(p, mean.data_ptr(), rgba.data_ptr(), side.cuda_stream, False)
    torch.cuda.synchronize()
    # (1) hipMemGetInfo before and after: nothing was allocated or freed on the device, with stats and without
    free0, _ = torch.cuda.mem_get_info(dev)
    s1 = ctx.render_device_noalloc(p, mean.data_ptr(), rgba.data_ptr(), None, True)
    ctx.render_device_noalloc(p, mean.data_ptr(), rgba.data_ptr(), side.cuda_stream, False)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info(dev)
    assert free1 == free0, (free0, free1)
    assert s1.batches >= 1 and torch.equal(rgba, want_rgba) and torch.equal(mean.view(torch.int32), want_mean.view(torch.int32))
    # (2) the enqueue-only mode (stats == NULL) inside a stream capture; the graph replayed twice
    mean.zero_(); rgba.zero_()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        ctx.render_device_noalloc(p, mean.data_ptr(), rgba.data_ptr(), torch.cuda.current_stream().cuda_stream, False)
    torch.cuda.synchronize()
    assert int(rgba.sum().item()) == 0                                  # capture enqueued nothing
    for _ in range(2):
        mean.zero_(); rgba.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(rgba, want_rgba) and torch.equal(mean.view(torch.int32), want_mean.view(torch.int32))
    del g
