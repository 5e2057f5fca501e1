"""The C-ABI library loads without a GPU, exports every symbol include/mi_tracer.h
declares, its structs have the layout the bindings assume, and compute entry points
fail loudly (status + message) rather than falling back when there is no device."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

import pathtracer_amd as P
from pathtracer_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mi_tracer.h")


def declared_functions():
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported():
    names = declared_functions()
    assert len(names) >= 11 and set(names) == set(N.MI_SYMBOLS)
    for n in names:
        assert hasattr(N.mi, n), f"{n} declared in mi_tracer.h but not exported"


def test_struct_layouts_match_the_header(tmp_path):
    prog = tmp_path / "sz.c"
    prog.write_text('#include <stdio.h>\n#include "mi_tracer.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(mi_cubemap), sizeof(mi_build_stats), sizeof(mi_scene_soa), '
                    'sizeof(mi_render_params), sizeof(mi_stats), offsetof(mi_scene_soa, cam_eye), offsetof(mi_render_params, batch_paths));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    assert [int(x) for x in out] == [C.sizeof(N.CubeMap), C.sizeof(N.BuildStats), C.sizeof(N.SceneSoA), C.sizeof(N.RenderParams), C.sizeof(N.Stats),
                                     N.SceneSoA.cam_eye.offset, N.RenderParams.batch_paths.offset]


def test_header_is_plain_c_and_cites_the_reference():
    src = open(HEADER).read()
    assert "extern \"C\"" in src and "torch" not in src.lower().replace("no torch types", "")
    for cite in ("tracer.h:19-52", "GpuTracer.cu:161-189", "bvh.cu:395-415", "scene.cu:3-13"):
        assert cite in src


def test_no_product_code_uses_the_oracle():
    for base, _, files in os.walk(os.path.join(ROOT, "pathtracer_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                text = open(os.path.join(base, f), errors="ignore").read()
                assert "oracle_py" not in text and "liboracle" not in text and "oracle/" not in text.replace("oracle/oracle.cpp", ""), f


def test_shard_pixels_and_param_validation_need_no_gpu():
    p = P.Context.params(80, 81, 1, 0, shard=(0, 3))
    assert P.shard_pixels(p) == 3 * 32 * 32          # 9 tiles of 32x32 over 3 shards
    assert P.shard_pixels(P.Context.params(80, 81, 1, 0, shard=(5, 3))) == -1
    assert P.shard_pixels(P.Context.params(0, 81, 1, 0)) == -1


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_without_a_device_everything_fails_loudly():
    with pytest.raises(P.MiError) as e:
        P.Context(0)
    assert e.value.status == N.MI_ERR_NO_DEVICE and "HIP device" in str(e.value)
    with pytest.raises(P.MiError):
        P.HipTracer()
    with pytest.raises(P.MiError) as e:
        P.Group([0])                      # no device: the group fails like the context, before RCCL is touched
    assert e.value.status == N.MI_ERR_NO_DEVICE
    with pytest.raises(P.MiError):
        P.HipTracer(n_devices=2)
    assert N.mi.mi_group_create(None, 0, None) == N.MI_ERR_INVALID
    assert N.mi.mi_group_render(None, None, None, None) == N.MI_ERR_INVALID
    assert N.mi.mi_render_device_noalloc(None, None, None, None, None, None) == N.MI_ERR_INVALID
    assert N.mi.mi_scene_upload(None, None) == N.MI_ERR_INVALID
    assert N.mi.mi_render(None, None, None, None, None) == N.MI_ERR_INVALID
    assert b"NULL" in N.mi.mi_last_error()


def test_cli_reports_missing_arguments():
    exe = os.path.join(ROOT, "pathtracer_amd", "bin", "ray")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "no input and/or output name" in r.stderr   # CommandLineUI.cu:70-73
