#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (oracle/oracle.cpp).

These are REGRESSION vectors of this repository's oracle, not outputs of the reference:
/root/reference cannot be compiled in this image (it needs CUDA headers and glm) and
ships no fixtures of its own, so the oracle stays "parity unpinned" against it
(DESIGN.md §Oracle).  The vectors let the GPU box check the HIP path against fixed
bytes even if the oracle library were stale, and catch accidental oracle changes.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import pathtracer_amd as P  # noqa: E402
from pathtracer_amd import scenes  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

CASES = {
    # name: (scene text, width, samples, depth, seed)
    "simple": (scenes.simple_ray(), 48, 2, 2, 1),
    "cornell": (scenes.cornell_ray(), 40, 2, 5, 2),
    "glass": (scenes.glass_ray(), 64, 1, 8, 3),
    "blob2": (scenes.blob_ray(2), 64, 1, 3, 4),
}


def main():
    for name, (text, width, samples, depth, seed) in CASES.items():
        sc = P.Scene.from_text(text)
        w, h = sc.image_size(width)
        ids, t, nrm, _ = O.trace_primary(sc, w, h, samples, O.BRUTE)
        mean, rgb, cnt = O.render(sc, P.Context.params(w, h, samples, depth, seed=seed))
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), text=np.array(text), size=np.array([w, h, samples, depth, seed]),
                            prim_id=ids, t=t, normal=nrm, rgb=rgb, mean=mean, rays=np.array([cnt.rays], np.uint64))
        print(name, w, h, "hits", int((ids >= 0).sum()), "rays", cnt.rays)
    hable = np.array([O.tonemap_byte(np.float32(k / 1024.0)) for k in range(1025)], np.uint8)
    np.savez_compressed(os.path.join(HERE, "tonemap.npz"), bytes=hable)


if __name__ == "__main__":
    main()
