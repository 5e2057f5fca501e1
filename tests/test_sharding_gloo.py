"""The N>1 path on CPU: two gloo ranks each render their tile shard, the shard-local
buffers are gathered to rank 0 and assembled.  The compute is done by the CPU oracle
here (no GPU in this container); the sharding layout, the gather and the assembly are
the code the GPU bench uses."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pathtracer_amd as P
    from pathtracer_amd import scenes, sharding
    from oracle import oracle_py as O
    sc = P.Scene.from_text(scenes.simple_ray())
    w, h = sc.image_size(80)
    p = P.Context.params(w, h, 1, 2, seed=9, shard=(rank, world))
    rgb = O.render(sc, p)[1]
    stride = sharding.shard_pixels(w, h, 0, world)           # rank 0 owns the most tiles
    assert stride == P.shard_pixels(P.Context.params(w, h, 1, 2, shard=(0, world)))
    local = np.zeros((stride, 3), np.uint8)
    mine = sharding.extract_local(rgb, rank, world)
    local[: len(mine)] = mine
    t = torch.from_numpy(local)
    bufs = [torch.zeros_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, bufs, dst=0)
    if rank == 0:
        frame = sharding.assemble([b.numpy() for b in bufs], w, h)
        full = O.render(sc, P.Context.params(w, h, 1, 2, seed=9))[1]
        np.save(out_path, np.array([np.array_equal(frame, full)]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_tile_sharding_reassembles_the_single_rank_image(tmp_path):
    out = str(tmp_path / "ok.npy")
    mp.spawn(_worker, args=(2, 29000 + os.getpid() % 2000, out), nprocs=2, join=True)
    assert np.load(out)[0]
