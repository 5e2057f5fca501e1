"""bench.py's roofline divides counters from a committed PMC pass by a time measured live; pmc_record() must refuse
the pass when it does not describe the kernels being timed (VERDICT r2 / ADVICE r2: the tie used to be discipline,
not construction).  CPU-only: a copy of the kernel sources + a PMC file in a temporary tree."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

WL = "C3-blob82k-1080p-16spp"


def make_tree(tmp_path):
    for rel in bench.KERNEL_SOURCES:
        dst = tmp_path / rel
        dst.parent.mkdir(parents=True, exist_ok=True)
        shutil.copy(os.path.join(ROOT, rel), dst)
    (tmp_path / "profiles").mkdir()
    rec = {WL: {"tag": "t", "source": "profiles/t_pmc.json", "kernel_source_sha256": bench.kernel_source_sha256(str(tmp_path)), "abi_version": 12,
                "rays_cast_frame": 1000, "extend_launches_per_frame": 7, "k_extend": {"lane_ops_sum": 1.0}}}
    (tmp_path / "profiles" / "roofline_pmc.json").write_text(json.dumps(rec))
    return str(tmp_path)


def test_matching_pass_is_used(tmp_path, monkeypatch):
    root = make_tree(tmp_path)
    rec, stale = bench.pmc_record(WL, 1, "host", 1000, root)
    assert rec is not None and stale is None and rec["k_extend"]["lane_ops_sum"] == 1.0
    assert bench.pmc_record(WL, 2, "host", None, root)[0] is not None          # a shard: no frame ray count to compare
    assert bench.pmc_record("C2-cornell-512-64spp", 1, "host", 5, root) == (None, None)   # never profiled: absent, not stale
    assert bench.pmc_record(WL, 1, "device", 1000, root)[0] is not None         # the device's SAH build is the host's tree
    monkeypatch.setenv("MI_TRACER_BUILDER", "ploc")
    assert bench.pmc_record(WL, 1, "device", 1000, root) == (None, None)       # another tree: no pass on file
    monkeypatch.delenv("MI_TRACER_BUILDER")


def test_kernel_edit_without_a_new_pass_is_stale(tmp_path):
    root = make_tree(tmp_path)
    with open(os.path.join(root, bench.KERNEL_SOURCES[0]), "a") as f:
        f.write("\n// an edit\n")
    rec, stale = bench.pmc_record(WL, 1, "host", 1000, root)
    assert rec is None and stale and "kernel sources" in stale


def test_other_frame_is_stale(tmp_path):
    root = make_tree(tmp_path)
    rec, stale = bench.pmc_record(WL, 1, "host", 1001, root)                   # other knobs / batch sizes / scene: other ray count
    assert rec is None and stale and "rays" in stale


def test_committed_pass_has_the_tie_fields():
    path = os.path.join(ROOT, "profiles", "roofline_pmc.json")
    d = json.load(open(path))
    for wl, rec in d.items():
        assert len(rec.get("kernel_source_sha256", "")) == 64, wl
