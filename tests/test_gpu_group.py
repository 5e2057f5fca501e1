"""mi_group: several GPUs in one process behind the Tracer interface (scene replicated, frame sharded by tile, RGBA8
tiles to the first device over RCCL, assembled there).  A lease has ONE GPU, so this runs the group path with
n = 1 — RCCL is loaded, the communicator created, the (empty) grouped exchange and the assembly executed — and
checks the frame against the plain context and the oracle; n > 1 is the same code and unverified on hardware."""
import ctypes as C
import subprocess

import numpy as np
import pytest

import pathtracer_amd as P
from pathtracer_amd import _native as N
from pathtracer_amd import scenes
from oracle import oracle_py as O

pytestmark = pytest.mark.gpu


def test_group_of_one_renders_the_same_frame(ctx, scenes_small):
    g = P.Group([0])
    assert len(g) == 1
    for name, width, s, d in (("cornell", 96, 2, 5), ("blob3", 200, 2, 3)):
        sc = scenes_small[name]
        w, h = sc.image_size(width)
        p = P.Context.params(w, h, s, d, seed=6)
        g.upload(sc)
        rgb, st = g.render(p)
        ctx.upload(sc)
        _, want, st0 = ctx.render(p, want_mean=False)
        assert np.array_equal(rgb, want) and st.rays_cast == st0.rays_cast and st.paths == st0.paths
        _, orgb, cnt = O.render(sc, p)
        assert np.array_equal(rgb, orgb) and st.rays_cast == cnt.rays
        b = g.upload_build(sc)                       # every device builds its own tree
        assert b.n_nodes == sc.n_prims - 1
        rgb2, _ = g.render(p)
        assert np.array_equal(rgb2, want)
    ps = P.Context.params(32, 32, 1, 1, shard=(0, 2))
    out = np.zeros((32, 32, 3), np.uint8)
    assert N.mi.mi_group_render(g._h, C.byref(ps), out.ctypes.data, None) == N.MI_ERR_INVALID   # the group shards by itself
    g.close()


def test_group_rejects_bad_device_lists():
    h = C.c_void_p()
    two = (C.c_int * 2)(0, 0)
    assert N.mi.mi_group_create(two, 2, C.byref(h)) == N.MI_ERR_INVALID and b"twice" in N.mi.mi_last_error()
    assert N.mi.mi_group_create(two, 0, C.byref(h)) == N.MI_ERR_INVALID
    far = (C.c_int * 1)(63)
    assert N.mi.mi_group_create(far, 1, C.byref(h)) == N.MI_ERR_NO_DEVICE


def test_tracer_and_cli_through_a_group(tmp_path, scenes_small):
    text = scenes.simple_ray()
    sc = scenes_small["simple"]
    w, h = sc.image_size(80)
    t = P.HipTracer(depth=2, samples=2, seed=5, n_devices=1)
    assert t.load_scene(text=text)
    t.trace_setup(w, h)
    t.trace_image(w, h)
    t.wait_render()
    _, orgb, _ = O.render(sc, P.Context.params(w, h, 2, 2, seed=5))
    assert np.array_equal(t.get_buffer(), orgb)
    import os
    ray = os.path.join(os.path.dirname(P.__file__), "bin", "ray")
    src, out = tmp_path / "s.ray", tmp_path / "o.bmp"
    src.write_text(text)
    r = subprocess.run([ray, "-r", "2", "-w", "80", "-s", "2", "-S", "5", "-N", "1", str(src), str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    img = P.read_image(out)
    assert img is not None and np.array_equal(img, orgb)


def test_cli_builds_the_tree_on_the_gpu_by_default(tmp_path, scenes_small):
    # pt::HipTracer builds the BVH on the GPU (TraceOptions::device_bvh = 1; -H: on the host; MI_TRACER_BUILDER picks the
    # device builder): same image every way, the oracle's
    import os
    text = scenes.blob_ray(3)
    sc = scenes_small["blob3"]
    w, h = sc.image_size(120)
    _, orgb, _ = O.render(sc, P.Context.params(w, h, 2, 3, seed=7))
    ray = os.path.join(os.path.dirname(P.__file__), "bin", "ray")
    src = tmp_path / "b.ray"
    src.write_text(text)
    for flags, env in (([], {}), (["-H"], {}), (["-B"], {}), ([], {"MI_TRACER_BUILDER": "ploc"})):
        out = tmp_path / "o.bmp"
        r = subprocess.run([ray, "-r", "3", "-w", "120", "-s", "2", "-S", "7", *flags, str(src), str(out)], capture_output=True, text=True, timeout=120,
                           env={**os.environ, **env})
        assert r.returncode == 0, r.stderr
        assert np.array_equal(P.read_image(out), orgb), (flags, env)
