import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """One mi_ctx for the whole GPU session (no fallback: fails if there is no device)."""
    import pathtracer_amd as P
    c = P.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def scenes_small():
    import pathtracer_amd as P
    from pathtracer_amd import scenes
    return {
        "simple": P.Scene.from_text(scenes.simple_ray()),
        "cornell": P.Scene.from_text(scenes.cornell_ray()),
        "glass": P.Scene.from_text(scenes.glass_ray()),
        "blob3": P.Scene.from_text(scenes.blob_ray(3)),
    }
