"""Host-side readImage (BMP/PNG) and matchCubemapFiles — counterparts of
src/fileio/images.cu:51-57, bitmap.cu:17-93, pngimage.cu:195-216 and TraceUI.cu:86-148.
CPU only."""
import numpy as np
import pytest
from PIL import Image

import pathtracer_amd as P


@pytest.mark.parametrize("size", [(5, 3), (16, 16), (1, 7), (33, 2)])
def test_bmp_and_png_readers_return_bottom_up_rgb(tmp_path, size):
    rng = np.random.default_rng(size[0] * 100 + size[1])
    w, h = size
    a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)                 # row 0 = TOP row, as PIL sees it
    Image.fromarray(a).save(tmp_path / "a.bmp")
    Image.fromarray(a).save(tmp_path / "a.png")
    Image.fromarray(a).save(tmp_path / "a0.png", compress_level=0)
    for name in ("a.bmp", "a.png", "a0.png"):
        got = P.read_image(tmp_path / name)
        assert got is not None and got.shape == (h, w, 3) and np.array_equal(got, a[::-1]), name
    # other PNG colour types: RGBA (alpha dropped), grey, grey+alpha, palette
    rgba = np.dstack([a, rng.integers(0, 256, (h, w), dtype=np.uint8)])
    Image.fromarray(rgba, "RGBA").save(tmp_path / "rgba.png")
    assert np.array_equal(P.read_image(tmp_path / "rgba.png"), a[::-1])
    Image.fromarray(a[..., 0], "L").save(tmp_path / "g.png")
    assert np.array_equal(P.read_image(tmp_path / "g.png"), np.repeat(a[::-1, :, :1], 3, axis=2))
    pal = Image.fromarray(a).quantize(16)
    pal.save(tmp_path / "p.png")
    got = P.read_image(tmp_path / "p.png")
    if got is not None:     # PIL may write < 8-bit palettes for few colours: those are (documented) unsupported
        assert np.array_equal(got, np.array(pal.convert("RGB"))[::-1])


def test_unreadable_images_are_refused(tmp_path):
    (tmp_path / "x.png").write_bytes(b"not a png")
    (tmp_path / "x.bmp").write_bytes(b"BM" + b"\0" * 60)
    (tmp_path / "x.jpg").write_bytes(b"\xff\xd8")
    for n in ("x.png", "x.bmp", "x.jpg", "missing.png"):
        assert P.read_image(tmp_path / n) is None
    a = np.zeros((4, 4, 3), np.uint8)
    Image.fromarray(a.astype(np.uint16)[..., 0] * 257).save(tmp_path / "d16.png")   # 16-bit grey
    assert P.read_image(tmp_path / "d16.png") is None


def test_match_cubemap_files(tmp_path):
    a = np.zeros((2, 2, 3), np.uint8)
    names = ["sky_posx.png", "sky_negx.png", "sky_posy.bmp", "sky_negy.png", "sky_posz.png", "sky_negz.png"]
    for n in names:
        Image.fromarray(a).save(tmp_path / n)
    (tmp_path / "readme.txt").write_text("posx")            # ignored: no image extension
    assert P.match_cubemap_files(tmp_path / "sky_negy.png") == names
    (tmp_path / "sky_negz.png").unlink()                   # only five faces
    assert P.match_cubemap_files(tmp_path / "sky_negy.png") is None
    # two candidates for +x are refused (the reference stops scanning once six names matched,
    # TraceUI.cu:138-139, so with five other faces the clash is always seen)
    Image.fromarray(a).save(tmp_path / "other_posx.png")
    assert P.match_cubemap_files(tmp_path / "sky_negy.png") is None
    assert P.match_cubemap_files("no_directory_part.png") is None
