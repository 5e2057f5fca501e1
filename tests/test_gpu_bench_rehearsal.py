"""The N > 1 line of bench.py, rehearsed so that it cannot rot before an 8-GPU node runs it: a fresh
`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 --backend gloo --check` with both ranks on the
one GPU of the box (tiles staged through the host, gloo gather; the RCCL exchange itself needs two GPUs and stays
unverified on hardware).  Asserts what the driver will rely on: one JSON line from rank 0 with the contract's keys,
`phases` per rank, and the assembled two-rank frame equal to the single-rank frame byte for byte (k_assemble).
No reference counterpart (GpuTracer.cu:165-181 is one device); sharding is north_star functionality."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_bench_line_on_one_gpu():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--check",
           "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "assembled 2-rank frame == single-rank frame: True" in r.stderr, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 prints ONE line
    out = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "phases"):
        assert k in out, k
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["value"] > 0
    ph = out["phases"]
    assert len(ph["shard_ms"]) == 2 and len(ph["rays_per_step_by_rank"]) == 2 and ph["gather_ms"] >= 0 and ph["assemble_ms"] >= 0
    # both shards carry rays, and together they are the frame's rays
    assert all(x > 0 for x in ph["rays_per_step_by_rank"])
    assert abs(sum(ph["rays_per_step_by_rank"]) - out["config"]["rays_per_step"]) < 1
