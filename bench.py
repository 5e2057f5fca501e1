#!/usr/bin/env python3
"""bench.py — Mrays/s of the HIP wavefront path tracer on the BASELINE workload.

A step = one full render of the workload image (every pixel, every sample, all
bounces), inputs (flattened scene) already resident in HBM.  With N ranks the image
is sharded by 32x32 tile (tile k -> rank k % N, BVH replicated); every rank renders
its tiles and the tonemapped tiles are gathered to rank 0 over RCCL and assembled
there, inside the timed region.  value = closest-hit queries of all ranks / time.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
VALU_PEAK_TFLOPS = 157.3     # fp32 vector peak (spec): 256 CUs x 4 SIMDs x 32 lanes/clock x 2 flop x 2.4 GHz
VALU_PEAK_TLANEOPS = VALU_PEAK_TFLOPS / 2.0   # lane operations: one full wave64 VALU instruction per SIMD every 2 cycles
L2_PEAK_GBS = 17000.0        # MI355X_MICROARCH.md: 16.8-18.8 TB/s chip-wide for rows served from the XCDs' L2s

WORKLOADS = {
    # name: (scene builder, width, samples/dim, depth)   — SURVEY §8d
    "C3-blob82k-1080p-16spp": (lambda S: S.blob_scene(6), 1920, 4, 5),
    "C2-cornell-512-64spp": (lambda S: __import__("pathtracer_amd").Scene.from_text(S.cornell_ray()), 512, 8, 5),
    "C5-glass-1024-16spp": (lambda S: __import__("pathtracer_amd").Scene.from_text(S.glass_ray()), 1024, 4, 8),
    "C4-blob1.3M-4k-256spp": (lambda S: S.blob_scene(8), 3840, 16, 5),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="C3-blob82k-1080p-16spp", choices=sorted(WORKLOADS))
    ap.add_argument("--bvh", default="device", choices=["host", "device"],
                    help="device (default): built by mi_scene_upload_build (top-down binned SAH; MI_TRACER_BUILDER=ploc | lbvh for the other builders); "
                         "host: the same tree built by libpt_host")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearsal of the N>1 path with several ranks on ONE GPU (tiles staged through the host); never a result")
    ap.add_argument("--check", action="store_true", help="N>1: rank 0 also renders the whole frame alone and compares the assembled frame byte for byte")
    ap.add_argument("--batch-paths", type=int, default=0)
    ap.add_argument("--streams", type=int, default=2, choices=[1, 2],
                    help="2 (default): the batches of a frame alternate between two streams; 1: one stream (per-kernel profiling runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sync-steps", action="store_true", help="wait for every frame on the host before the next is enqueued (round 2's loop)")
    ap.add_argument("--no-overlap", action="store_true", help="N>1: wait for every step's exchange before the next render starts (default: the exchange of step i runs beside the render of step i+1)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import pathtracer_amd as P
    from pathtracer_amd import scenes as S

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if args.backend == "gloo":
        local_rank = 0  # rehearsal: every rank on the one GPU of the box
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    build, width, samples, depth = WORKLOADS[args.workload]
    scene = build(S)
    w, h = scene.image_size(width)
    ctx = P.Context(local_rank)
    build = ctx.upload_build(scene) if args.bvh == "device" else None  # scene resident in HBM before any timed region
    if build is None:
        ctx.upload(scene)
    sah_cost = ctx.scene_quality()

    params = P.Context.params(w, h, samples, depth, seed=0, shard=(rank, world), batch_paths=args.batch_paths, one_stream=args.streams == 1)
    # the timed loop only ENQUEUES its frames (MI_RENDER_ENQUEUE_ONLY): the GPU goes from one frame to the next without a host
    # round trip, the fence behind the K steps is the one wait; --sync-steps waits for every frame as round 2 did
    params_q = P.Context.params(w, h, samples, depth, seed=0, shard=(rank, world), batch_paths=args.batch_paths, one_stream=args.streams == 1, enqueue_only=True)
    p0 = P.Context.params(w, h, samples, depth, seed=0, shard=(0, world))
    stride = P.shard_pixels(p0)  # rank 0 owns the most tiles
    # Two output buffers per rank: step i renders into buffer i % 2, so that for N > 1 the exchange of step i (RCCL's own
    # stream) runs beside the render of step i + 1 — a shard of a 6-ms frame renders in 1.5-3 ms, the gather of its tiles
    # and the assembly are 0.1-0.3 ms that need not be serial.
    rgba8_bufs = [torch.zeros(stride * 4, dtype=torch.uint8, device=dev) for _ in range(2 if world > 1 else 1)]
    rgba8 = rgba8_bufs[0]
    gathered_bufs = [torch.zeros(world * stride * 4, dtype=torch.uint8, device=dev) for _ in range(2)] if (world > 1 and rank == 0) else None
    gathered = gathered_bufs[0] if gathered_bufs else None
    frame = torch.zeros(w * h * 3, dtype=torch.uint8, device=dev) if (world > 1 and rank == 0) else None  # assembled frame stays in HBM, like N=1
    # One explicit (non-default) torch stream carries the renders, the collective's stream dependencies and the
    # assembly: torch's default stream has handle 0, which mi_render_device would read as "the context's own
    # stream", ordered with neither torch's current stream nor RCCL's.
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream

    # Exchange step: RGBA8 tiles of every rank -> rank 0.  dist.gather (grouped send/recv in RCCL) moves
    # only what rank 0 needs; should the backend refuse it, all ranks fall back together to an all-gather.
    use_gather = True
    if world > 1 and args.backend == "gloo":
        host_local = [torch.zeros(stride * 4, dtype=torch.uint8) for _ in range(2)]
        host_all = [[torch.zeros(stride * 4, dtype=torch.uint8) for _ in range(world)] for _ in range(2)] if rank == 0 else [None, None]
    if world > 1 and args.backend == "nccl":
        ok = torch.ones(1, device=dev)
        try:
            dist.gather(rgba8, list(gathered.view(world, -1).unbind(0)) if rank == 0 else None, dst=0)
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            print(f"[bench] rank {rank}: dist.gather unavailable ({e}); using all_gather_into_tensor", file=sys.stderr)
            ok.zero_()
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        use_gather = bool(ok.item())
        if not use_gather and gathered_bufs is None:
            gathered_bufs = [torch.zeros(world * stride * 4, dtype=torch.uint8, device=dev) for _ in range(2)]

    def exchange_start(b):
        """Starts the exchange of output buffer b (asynchronously unless --no-overlap); returns the handle finish() takes."""
        g = gathered_bufs[b] if gathered_bufs else None
        if args.backend == "gloo":
            host_local[b].copy_(rgba8_bufs[b])          # rehearsal: tiles staged through the host
            return dist.gather(host_local[b], host_all[b], dst=0, async_op=True)
        if use_gather:
            return dist.gather(rgba8_bufs[b], list(g.view(world, -1).unbind(0)) if rank == 0 else None, dst=0, async_op=True)
        return dist.all_gather_into_tensor(g, rgba8_bufs[b], async_op=True)

    def exchange_finish(work, b):
        """Waits for the exchange of buffer b on every rank (its send reads rgba8_bufs[b], which the render after next
        overwrites) and assembles the frame on rank 0."""
        work.wait()                                      # nccl: the current stream waits (the renders enqueued behind it are ordered with it); gloo: the host does
        if rank == 0:
            if args.backend == "gloo":
                gathered_bufs[b].copy_(torch.cat(host_all[b]))
            ctx.assemble_rgb8_device(params, gathered_bufs[b].data_ptr(), stride, frame.data_ptr(), stream)
            return frame
        return None

    pending = []   # [(work, buffer)] exchanges in flight: at most one with the overlap, none without
    step_no = [0]

    def step(wait=False):
        b = step_no[0] % len(rgba8_bufs)
        step_no[0] += 1
        # wait: returns when the shard is rendered, with its stats; otherwise the frame is enqueued on `stream` (None returned)
        st = ctx.render_device(params if (wait or args.sync_steps) else params_q, None, rgba8_bufs[b].data_ptr(), stream)
        img = None
        if world > 1:
            while pending:                               # the exchange of the step before ran beside this render
                img = exchange_finish(*pending.pop(0))
            work = exchange_start(b)
            if args.no_overlap:
                img = exchange_finish(work, b)
            else:
                pending.append((work, b))
        return st, img

    def drain():
        img = None
        while pending:
            img = exchange_finish(*pending.pop(0))
        return img

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ref, _ = step(wait=True)   # untimed: what one step of this rank casts (the image, and with it the ray count, is the same every step)
    drain()
    fence()
    t0 = time.perf_counter()
    img = None
    for _ in range(args.steps):
        st, im = step()
        img = im if im is not None else img
        if st is not None and st.rays_cast != ref.rays_cast:
            raise SystemExit(f"step cast {st.rays_cast} rays, the reference step {ref.rays_cast}")
    rays = ref.rays_cast * args.steps
    rays_ext = ref.rays_extend * args.steps
    ms_extend, n_ext = 0.0, 0
    last = drain()                                       # the last step's exchange and assembly belong to the timed region
    img = last if last is not None else img
    fence()
    dt = time.perf_counter() - t0

    tot = torch.tensor([float(rays), dt, ms_extend, float(n_ext)], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
    if world > 1:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tot.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        dt_max, rays_all = mx[1].item(), sm[0].item()
    else:
        dt_max, rays_all = dt, float(rays)

    if rank == 0 and world > 1 and args.check:
        pf = P.Context.params(w, h, samples, depth, seed=0)
        solo = ctx.render(pf, want_mean=False)[1]
        same = bool((img.cpu().numpy().reshape(h, w, 3) == solo).all())
        print(f"[bench] assembled {world}-rank frame == single-rank frame: {same}", file=sys.stderr)
        if not same:
            raise SystemExit("assembled frame differs from the single-rank render")
    # N > 1: where a step's time goes on every rank — three more steps OUTSIDE the timed region with a host sync
    # after each phase (render of the shard | gather | assembly on rank 0)
    phases = None
    if world > 1:
        acc = [0.0, 0.0, 0.0]
        for _ in range(3):
            torch.cuda.synchronize(); dist.barrier()
            t_a = time.perf_counter()
            ctx.render_device(params, None, rgba8.data_ptr(), stream)
            t_b = time.perf_counter()
            wk = exchange_start(0)
            wk.wait()
            torch.cuda.synchronize()
            t_c = time.perf_counter()
            if rank == 0:
                if args.backend == "gloo":
                    gathered_bufs[0].copy_(torch.cat(host_all[0]))
                ctx.assemble_rgb8_device(params, gathered_bufs[0].data_ptr(), stride, frame.data_ptr(), stream)
            t_d = time.perf_counter()
            acc = [acc[0] + (t_b - t_a) * 1e3 / 3, acc[1] + (t_c - t_b) * 1e3 / 3, acc[2] + (t_d - t_c) * 1e3 / 3]
        mine = torch.tensor(acc + [float(rays) / args.steps], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        allp = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allp, mine)
        allp = torch.stack(allp).cpu().numpy()
        phases = {"shard_ms": [float(x) for x in allp[:, 0]], "shard_ms_max": float(allp[:, 0].max()), "shard_ms_min": float(allp[:, 0].min()),
                  "gather_ms": float(allp[0, 1]), "gather_ms_by_rank": [float(x) for x in allp[:, 1]], "assemble_ms": float(allp[0, 2]),
                  "rays_per_step_by_rank": [float(x) for x in allp[:, 3]],
                  "note": "3 extra steps outside the timed region, host sync after each phase; gather_ms on rank 0 includes waiting for the slowest shard"}
    if rank == 0:
        # Roofline of the dominant kernel (k_extend).  What binds it is VALU issue with partly filled waves (DESIGN.md
        # section 3): achieved = active-lane VALU operations per second = SQ_THREAD_CYCLES_VALU of one frame's k_extend
        # launches (a property of code + workload, from the committed PMC pass, profiles/roofline_pmc.json) / the time
        # of those launches measured LIVE here with HIP events; peak = every SIMD issuing a full wave64 instruction
        # every 2 cycles = the guide's fp32 vector peak / 2 flop.  The algorithmic byte rate of SURVEY 8(d) is kept
        # beside it as algorithmic_gbs (the scene is cache resident: not a roof that binds).
        pc = P.Context.params(w, h, samples, depth, seed=0, shard=(rank, world), batch_paths=args.batch_paths, count_work=True)
        stc = ctx.render_device(pc, None, rgba8.data_ptr(), stream)
        r = max(1, stc.rays_cast)
        nodes_per_ray, leaves_per_ray = stc.node_visits / r, stc.prim_tests / r
        # SURVEY 8d prices a visit of a binary node (2 boxes, 64 B, 54 flops); the kernels walk 4-wide nodes
        # collapsed from the binary tree: 4 boxes = 108 flops, and 64 B, because the boxes are 16-bit codes
        bytes_per_ray = 64.0 * nodes_per_ray + 64.0 * leaves_per_ray + 128.0
        flops_per_ray = 108.0 * nodes_per_ray + 37.0 * leaves_per_ray + 160.0
        rays_rank0_per_step = rays_ext / args.steps   # rays traced by k_extend launches (k_tail's excluded)
        # Kernel time.  In the timed region the batches of a frame run on two streams and their launches overlap: a launch's
        # own duration (HIP events on its stream; what rocprofv3 --kernel-trace shows for this same command) then includes
        # time in which it shares the chip, and the durations of a step add up to more than the step.  The roofline of the
        # KERNEL is therefore taken from `serial_steps` more steps of the same workload, live, in this process, with every
        # batch on one stream (MI_RENDER_ONE_STREAM) — each launch alone on the chip; the overlapped figures stay beside it.
        # The timed region records no per-launch events (they cost ~2 % of a C3 step); the launches' own durations under
        # overlap come from `serial_steps` more steps as they run in the timed region, with the events on.
        serial_steps = max(3, args.steps // 2)
        ps2 = P.Context.params(w, h, samples, depth, seed=0, shard=(rank, world), batch_paths=args.batch_paths, one_stream=args.streams == 1, launch_times=True)
        ctx.render_device(ps2, None, rgba8.data_ptr(), stream)
        o_ms_ext = o_ms_tot = 0.0
        o_n = 0
        for _ in range(serial_steps):
            s2 = ctx.render_device(ps2, None, rgba8.data_ptr(), stream)
            o_ms_ext += s2.ms_extend; o_ms_tot += s2.ms_total; o_n += s2.extend_launches
        ovl_launches_per_step = max(1, o_n // serial_steps)
        ovl_avg_launch_ms = o_ms_ext / max(1, o_n)
        ovl_extend_ms_per_step = o_ms_ext / serial_steps
        ps1 = P.Context.params(w, h, samples, depth, seed=0, shard=(rank, world), batch_paths=args.batch_paths, one_stream=True, launch_times=True)
        ctx.render_device(ps1, None, rgba8.data_ptr(), stream)
        s_ms_ext = s_ms_tot = 0.0
        s_n = s_rays_ext = 0
        for _ in range(serial_steps):
            s1 = ctx.render_device(ps1, None, rgba8.data_ptr(), stream)
            s_ms_ext += s1.ms_extend; s_ms_tot += s1.ms_total; s_n += s1.extend_launches; s_rays_ext += s1.rays_extend
        rays_rank0_per_step = s_rays_ext / serial_steps
        launches_per_step = max(1, s_n // serial_steps)
        avg_launch_ms = s_ms_ext / max(1, s_n)
        extend_s_per_step = s_ms_ext / serial_steps * 1e-3
        algorithmic_gbs = bytes_per_ray * rays_rank0_per_step / extend_s_per_step / 1e9 if extend_s_per_step > 0 else 0.0
        valu_tflops = flops_per_ray * rays_rank0_per_step / extend_s_per_step / 1e12 if extend_s_per_step > 0 else 0.0
        frame_rays = stc.rays_cast if world == 1 else None      # N > 1: a rank's shard is scaled below; the source hash still binds
        pmc, pmc_stale = pmc_record(args.workload, world, args.bvh, frame_rays)
        if pmc and pmc.get("abi_version") not in (None, P._native.mi.mi_abi_version()):
            pmc, pmc_stale = None, f"{pmc.get('source')}: ABI {pmc.get('abi_version')}, this library {P._native.mi.mi_abi_version()}"
        roof = {"bound": "valu_issue", "kernel": "k_extend", "achieved": None, "peak": VALU_PEAK_TLANEOPS, "unit": "Tlane-op/s", "frac": None,
                "traffic": None, "avg_launch_ms": avg_launch_ms, "launches_per_step": launches_per_step,
                "extend_ms_per_step": extend_s_per_step * 1e3, "serial_ms_per_step": s_ms_tot / serial_steps, "serial_steps": serial_steps,
                "extend_share_of_serial_step": (s_ms_ext / serial_steps) / max(1e-9, s_ms_tot / serial_steps),
                "timed_region": {"launches_per_step": ovl_launches_per_step, "avg_launch_ms": ovl_avg_launch_ms, "extend_ms_per_step": ovl_extend_ms_per_step,
                                 "ms_per_step": o_ms_tot / serial_steps, "steps": serial_steps,
                                 "note": "launches as they run in the timed region (two streams, neighbouring batches overlap), measured on "
                                         "`steps` more steps with an event around every launch (MI_RENDER_LAUNCH_TIMES; the timed region itself "
                                         "records none, they cost ~2 % of a step): the launches' own durations, as rocprofv3 --kernel-trace of "
                                         "this command reports them; they sum to more than the step"},
                "rays_in_kernel_per_step": rays_rank0_per_step,
                "algorithmic_gbs": algorithmic_gbs, "algorithmic_hbm_frac": algorithmic_gbs / HBM_PEAK_GBS, "bytes_per_ray": bytes_per_ray,
                "nodes_per_ray": nodes_per_ray, "prim_tests_per_ray": leaves_per_ray,
                "valu_tflops": valu_tflops, "valu_flop_frac": valu_tflops / VALU_PEAK_TFLOPS,
                "note": "bound = VALU issue: frac = active-lane VALU operations per second / (1024 SIMDs x 32 lanes per clock x 2.4 GHz = fp32 vector "
                        "peak 157.3 TFLOP/s / 2); lane operations per frame from the committed PMC pass (SQ_THREAD_CYCLES_VALU), time live from "
                        "HIP events over serial_steps steps with every batch on ONE stream (each launch alone on the chip; in the timed region "
                        "the launches of two streams overlap, see timed_region and frac_timed_region; step_valu_frac is the whole step).  frac_4cycle_slots is the same product priced at one wave64 instruction per 4 cycles per SIMD with the "
                        "clock the PMC pass measured (VALU instruction rate x active lanes / 64).  algorithmic_gbs = SURVEY 8(d) bytes / "
                        "k_extend time: the scene is L1/L2/LDS resident, so that figure is not a roof that binds (hbm_counter_frac is)."}
        if pmc and world > 1:
            # a shard of the profiled frame: counter sums scaled by this rank's share of the frame's rays (the PMC pass is of
            # the whole frame on one GPU; lane operations per ray are a property of code + workload)
            share = (rays / args.steps) / pmc["rays_cast_frame"] if pmc.get("rays_cast_frame") else None
            if share:
                pmc = dict(pmc)
                for kname in ("k_extend", "k_shade", "k_tail"):
                    if pmc.get(kname):
                        pmc[kname] = {k: (v * share if k.endswith("_sum") and k != "duration_ms_sum" else v) for k, v in pmc[kname].items()}
                pmc["source"] = f"{pmc['source']} scaled by rank 0's share of the frame's rays ({share:.4f})"
            else:
                pmc = None
        if pmc:
            ke = pmc["k_extend"]
            # per k_extend launch, like `achieved`: counters of the launches that did work / launches per frame
            n_l = max(1, pmc.get("extend_launches_per_frame", launches_per_step))
            lane_ops_per_s = ke["lane_ops_sum"] / extend_s_per_step if extend_s_per_step > 0 else 0.0
            roof.update({"pmc_stale": False, "achieved": lane_ops_per_s / 1e12, "frac": lane_ops_per_s / 1e12 / VALU_PEAK_TLANEOPS,
                         "traffic": ke["hbm_bytes_sum"] / n_l,
                         "frac_4cycle_slots": ke["useful_lane_slots_4cyc"], "valu_inst_rate_4cycle": ke["valu_inst_rate"],
                         "lanes_active_of_64": ke["lanes_active"],
                         "hbm_counter_gbs": ke["hbm_bytes_sum"] / extend_s_per_step / 1e9, "hbm_counter_frac": ke["hbm_bytes_sum"] / extend_s_per_step / 1e9 / HBM_PEAK_GBS,
                         "l2_read_gbs": ke["l2_read_bytes_sum"] / extend_s_per_step / 1e9, "l2_read_frac": ke["l2_read_bytes_sum"] / extend_s_per_step / 1e9 / L2_PEAK_GBS,
                         "l2_hit": ke["l2_hit"], "l1_hit": ke["l1_hit"], "pmc_clock_ghz": ke.get("eff_clock_ghz"),
                         # the whole step, overlap included: active-lane VALU operations of every render kernel of a frame / the timed step
                         "step_valu_frac": sum((pmc.get(k) or {}).get("lane_ops_sum", 0.0) for k in ("k_extend", "k_shade", "k_tail"))
                                           / (dt_max / args.steps) / 1e12 / VALU_PEAK_TLANEOPS,
                         "frac_timed_region": ke["lane_ops_sum"] / (ovl_extend_ms_per_step * 1e-3) / 1e12 / VALU_PEAK_TLANEOPS if ovl_extend_ms_per_step > 0 else None,
                         "pmc": {"source": pmc["source"], "k_extend_by_level": ke["by_level"],
                                 "k_shade": {k: v for k, v in (pmc.get("k_shade") or {}).items() if k != "by_level"},
                                 "k_tail": {k: v for k, v in (pmc.get("k_tail") or {}).items() if k != "by_level"}}})
        elif pmc_stale:
            roof["pmc_stale"] = True
            roof["pmc"] = {"source": None, "stale": pmc_stale, "note": "the committed PMC pass does not describe the kernels being timed: achieved / frac / traffic "
                           "are withheld; run profiles/tools/collect.sh on this tree"}
        else:
            roof["pmc"] = {"source": None, "note": f"no PMC pass on file for workload {args.workload} / bvh {args.bvh} / {world} rank(s): run profiles/tools/collect.sh"}
        out = {
            "metric": "Mrays/s (whole node), bunny-class scene 1080p",
            "value": rays_all / dt_max / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": args.workload, "image": [w, h], "samples_per_dim": samples, "depth": depth,
                       "primitives": scene.n_prims,
                       "bvh": ("device " + {"sah": "binned SAH", "ploc": "PLOC", "lbvh": "LBVH"}[device_builder()]) if build else "host binned SAH", "bvh_depth": build.depth if build else scene.bvh_depth,
                       "bvh_sah_cost": sah_cost, "bvh_build_ms": build.ms_build if build else None,
                       "sharding": f"image tiles 32x32, tile k -> rank k % {world}, BVH replicated; sample-range sharding + sum-reduce not built "
                                   "(a reduce over ranks would round differently from the reference's per-pixel sample order)",
                       "exchange": None if world == 1 else (("RCCL gather of RGBA8 tiles to rank 0" if args.backend == "nccl" else "REHEARSAL: gloo, all ranks on one GPU")
                                                            + ("" if args.no_overlap else "; the exchange of step i overlaps the render of step i+1 (two output buffers)")),
                       "rays_per_step": rays_all / args.steps, "streams": args.streams,
                       "steps_enqueued_without_host_wait": not args.sync_steps},
            "roofline": roof,
        }
        if phases:
            out["phases"] = phases
        if world == 1 and not args.sync_steps:
            # the same K steps with a host wait after every frame (what a caller that needs each frame's stats pays)
            fence()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                ctx.render_device(params, None, rgba8.data_ptr(), stream)
            fence()
            out["value_sync_each_step"] = {"value": rays_all / (time.perf_counter() - t1) / 1e6, "unit": "Mrays/s",
                                           "what": "mi_render_device waits for every frame before the next is enqueued; `value` enqueues the K frames "
                                                   "(MI_RENDER_ENQUEUE_ONLY) and waits once, at the fence that closes the timed region"}
        if world == 1:
            # the Tracer boundary hands back HOST bytes (HipTracer::traceImage -> mi_render): the same frame with the RGB8 copy to a
            # pageable host buffer, PCIe-inclusive; never `value`
            pf = P.Context.params(w, h, samples, depth, seed=0, one_stream=args.streams == 1)
            n_h = max(3, args.steps // 2)

            def host_rate(buf):
                ctx.render(pf, want_mean=False, rgb_out=buf)
                t1 = time.perf_counter()
                rh = 0
                for _ in range(n_h):
                    rh += ctx.render(pf, want_mean=False, rgb_out=buf)[2].rays_cast
                return rh / (time.perf_counter() - t1) / 1e6

            # as the Tracer does it: ONE frame buffer (tracer.cu:87-93) that every traceImage renders into; pt::HipTracer keeps it
            # in pinned host memory (frameAlloc -> mi_host_alloc)
            pageable = host_rate(np.zeros((h, w, 3), np.uint8))
            pf_buf = P.PinnedFrame(h, w)
            pinned = host_rate(pf_buf.array)
            pf_buf.close()
            out["value_host_output"] = {"value": pinned, "unit": "Mrays/s", "steps": n_h, "frac_of_value": pinned / (rays_all / dt_max / 1e6),
                                        "pageable": pageable,
                                        "what": "mi_render with the RGB8 frame copied to the Tracer's host frame buffer (what HipTracer::traceImage calls), "
                                                "PCIe-inclusive; `value`: the buffer in pinned host memory (mi_host_alloc), where pt::HipTracer keeps it; "
                                                "`pageable`: the same into an ordinary buffer"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], band = cpu_baseline(scene, w, h, samples, depth, args.cpu_seconds)
            # SURVEY 8(d) "B2": what a user of the reference's own CpuTracer gets on a mesh of this class — its source cannot
            # travel to the GPU box, so the figure is the survey container's, quoted and labelled as such
            out["cpu_baseline"]["reference_as_shipped"] = {
                "value": 0.060, "unit": "Mrays/s/thread", "where": "survey container, 8 vCPU, g++ -O2, primary rays 640x360 on the 81,933-primitive mesh, "
                "reference BVH (genCluster, bvh.cu:66-145)", "src": "SURVEY.md section 6 / 8(d) B2 [measured there, not on this box]"}
            # second half of the metric ("RMSE vs CpuTracer"): the rows the oracle has just rendered against the
            # same rows of a GPU frame (raycheck's RMS over uint8 RGB, float subtraction); outside the timed region
            y0, y1, omean, orgb = band
            gmean, grgb, _ = ctx.render(P.Context.params(w, h, samples, depth, seed=0))
            d = grgb[y0:y1].astype(np.float64) - orgb[y0:y1].astype(np.float64)
            out["parity"] = {"rmse_rgb8_vs_oracle": float(np.sqrt((d * d).mean())) if d.size else 0.0,
                             "radiance_bit_identical": bool(np.array_equal(gmean[y0:y1].view(np.uint32), omean[y0:y1].view(np.uint32))),
                             "rows": [int(y0), int(y1)], "oracle": "CPU restatement of CpuTracer (parity unpinned, DESIGN.md section 5)"}
            from oracle import oracle_py as O
            pw, ph = scene.image_size(max(64, width // 4))   # primary hits of a quarter-size frame, 1 sample per pixel
            gid, gt, _ = ctx.trace_primary(pw, ph, 1)
            oid, ot, _, _ = O.trace_primary(scene, pw, ph, 1, O.BVH_CANONICAL)
            out["parity"].update({"primary_hit_id_agreement": float((gid == oid).mean()), "primary_hits": int((oid >= 0).sum()),
                                  "hit_t_bit_identical": bool(np.array_equal(gt.view(np.uint32), ot.view(np.uint32)))})
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


KERNEL_SOURCES = ("pathtracer_amd/csrc/mi_kernels.h", "pathtracer_amd/csrc/mi_tracer.hip", "pathtracer_amd/csrc/mi_bvh_build.h", "Makefile")  # the Makefile for its compiler flags


def device_builder():
    return os.environ.get("MI_TRACER_BUILDER") or "sah"


def kernel_source_sha256(root=ROOT):
    """sha256 over the device sources the render kernels are compiled from, in KERNEL_SOURCES order.  pmc_to_json.py
    stores it with every PMC pass; pmc_record() recomputes it, so counters of OTHER code are never divided by this
    code's time."""
    import hashlib
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(root, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def pmc_record(workload, world, bvh, rays_cast_frame=None, root=ROOT):
    """Counter-derived figures of the render kernels for this workload from the PMC passes committed under profiles/
    (profiles/roofline_pmc.json, written by profiles/tools/pmc_to_json.py from raw rocprofv3 --pmc output; host-built
    tree, the whole frame on one GPU — the caller scales it for a shard).  Returns (record | None, why_stale | None):
    the record is refused — and the roofline's counter-derived fields stay null, with `pmc_stale: true` — when the
    pass was taken from other kernel sources than the ones in this tree (sha256, kernel_source_sha256), with another
    ABI version, or when the frame it profiled cast another number of rays than this run's frame (other tuning
    knobs, batch sizes or scene)."""
    path = os.path.join(root, "profiles", "roofline_pmc.json")
    if (bvh != "host" and device_builder() != "sah") or not os.path.exists(path): # the device's SAH build is the host's tree; PLOC / LBVH trees are not
        return None, None
    with open(path) as f:
        rec = json.load(f).get(workload)
    if not (rec and rec.get("k_extend")):
        return None, None
    live = kernel_source_sha256(root)
    if rec.get("kernel_source_sha256") != live:
        return None, f"{rec.get('source')}: taken from kernel sources {str(rec.get('kernel_source_sha256'))[:16]}, this tree has {live[:16]}"
    if rays_cast_frame is not None and rec.get("rays_cast_frame") and int(rec["rays_cast_frame"]) != int(rays_cast_frame):
        return None, f"{rec.get('source')}: its frame cast {rec['rays_cast_frame']} rays, this run's frame {int(rays_cast_frame)}"
    return rec, None


def cpu_baseline(scene, w, h, samples, depth, budget_s):
    """The oracle (CPU restatement of CpuTracer, same flattened BVH, same RNG) timed on
    this host's cores over a bounded band of rows of the same workload."""
    import pathtracer_amd as P
    from oracle import oracle_py as O
    # threads = hardware_concurrency() in the reference (src/main.cu:5).  On the GPU box the visible CPUs are the whole
    # machine's while a one-GPU job's share is OMP_NUM_THREADS (16) of them: that many threads are used, both are stated.
    nproc = len(os.sched_getaffinity(0))
    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    cores = max(1, min(nproc, share) if share > 0 else nproc)
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "unknown")
    except OSError:
        pass
    rows = 2
    mid = h // 2
    p = P.Context.params(w, h, samples, depth, seed=0)
    # calibrate on 2 rows, then size the band for ~budget_s
    _, _, c = O.render(scene, p, threads=cores, rows=(mid, mid + rows))
    per_row = max(c.ms / rows, 1e-3)
    rows = int(max(2, min(h, budget_s * 1e3 / per_row)))
    y0 = max(0, mid - rows // 2)
    y1 = min(h, y0 + rows)
    omean, orgb, c = O.render(scene, p, threads=cores, rows=(y0, y1))
    return {"value": c.rays / (c.ms * 1e-3) / 1e6, "unit": "Mrays/s", "cores": cores, "host_nproc": nproc, "cpu_model": model, "kind": "port",
            "sample": f"rows [{y0},{min(h, y0 + rows)}) of the same {w}x{h} workload, {c.rays} rays in {c.ms / 1e3:.1f} s",
            "nodes_per_ray": c.node_visits / max(1, c.rays), "prim_tests_per_ray": c.leaf_tests / max(1, c.rays)}, (y0, y1, omean, orgb)


if __name__ == "__main__":
    main()
