"""Committed regression vectors (tests/golden, made by make_golden.py from the oracle).
CPU: the oracle still reproduces them.  GPU: the HIP path reproduces them byte for byte
through the C-ABI, independently of the oracle library on the box."""
import os

import numpy as np
import pytest

import pathtracer_amd as P
from oracle import oracle_py as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = ["simple", "cornell", "glass", "blob2"]


def load(name):
    g = np.load(os.path.join(GOLD, f"{name}.npz"))
    return g, P.Scene.from_text(str(g["text"])), [int(x) for x in g["size"]]


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_golden(name):
    g, sc, (w, h, samples, depth, seed) = load(name)
    ids, t, nrm, _ = O.trace_primary(sc, w, h, samples, O.BVH_CANONICAL)
    assert np.array_equal(ids, g["prim_id"]) and np.array_equal(t.view(np.uint32), g["t"].view(np.uint32))
    mean, rgb, cnt = O.render(sc, P.Context.params(w, h, samples, depth, seed=seed))
    assert np.array_equal(rgb, g["rgb"]) and cnt.rays == int(g["rays"][0])
    assert np.array_equal(mean.view(np.uint32), g["mean"].view(np.uint32))


def test_tonemap_golden():
    want = np.load(os.path.join(GOLD, "tonemap.npz"))["bytes"]
    assert [O.tonemap_byte(np.float32(k / 1024.0)) for k in range(1025)] == want.tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_reproduces_golden(ctx, name):
    g, sc, (w, h, samples, depth, seed) = load(name)
    ctx.upload(sc)
    ids, t, nrm = ctx.trace_primary(w, h, samples)
    assert np.array_equal(ids, g["prim_id"])
    assert np.array_equal(t.view(np.uint32), g["t"].view(np.uint32))
    assert np.array_equal(nrm.view(np.uint32), g["normal"].view(np.uint32))
    mean, rgb, st = ctx.render(P.Context.params(w, h, samples, depth, seed=seed))
    assert st.rays_cast == int(g["rays"][0])
    assert np.array_equal(rgb, g["rgb"]) and np.array_equal(mean.view(np.uint32), g["mean"].view(np.uint32))
