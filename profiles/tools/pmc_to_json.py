"""Raw rocprofv3 output of profiles/tools/collect.sh -> profiles/<tag>_pmc.json and the entry of this workload in
profiles/roofline_pmc.json (what bench.py reads for its `roofline` object).

usage: python profiles/tools/pmc_to_json.py <raw dir> <tag> <workload> [variant]
(`variant`, e.g. "wide8": a pass over another build of the library — profiles/<tag>_pmc.json is written, the roofline
file bench.py reads is left alone)

Every figure can be recomputed by hand from the committed JSON (raw counters per dispatch are kept):
  dispatches of one frame, in launch order, per batch and level k: k_tail (k >= 1) | k_extend | k_shade; counters and
  durations of a level are summed over the batches of the frame (one stream: `one_frame.py`, `bench.py --streams 1`).
  valu_inst_rate   = 4 * SQ_ACTIVE_INST_VALU / SIMD-cycles      (SQ_ACTIVE_INST_* count quad-cycles; SIMD-cycles =
                     SQ_BUSY_CYCLES / n_SE * n_SIMD: SQ_BUSY_CYCLES is summed over the 32 shader engines)
  lanes_active     = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU     (of 64)
  useful_lane_slots_4cyc = valu_inst_rate * lanes_active / 64         (VERDICT r1's definition: 4-cycle issue slots)
  lane_ops         = SQ_THREAD_CYCLES_VALU                             (active lanes summed over VALU instructions)
  hbm_bytes        = (2 * FETCH_SIZE + WRITE_SIZE) * 1024              (KiB; FETCH_SIZE doubled on gfx950, MI355X_MICROARCH.md)
  l2_read_bytes    = TCP_TCC_READ_REQ_sum * 64                         (requests of 64 B; an upper bound is 128 B each)
  l2_hit           = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum);  l1_hit = 1 - TCP_TCC_READ_REQ_sum / TCP_TOTAL_CACHE_ACCESSES_sum
  eff_clock_ghz    = GRBM_GUI_ACTIVE / 8 / duration                    (sum over the 8 XCDs)
Durations are the per-level medians of the same kernels in the --kernel-trace run of bench.py (13 frames)."""
import collections
import csv
import glob
import json
import os
import sys

N_SE, N_SIMD = 32, 1024


def counter_rows(d):
    fs = glob.glob(f"{d}/*/*counter_collection.csv") + glob.glob(f"{d}/*counter_collection.csv")
    if not fs:
        return {}
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        key = int(r["Dispatch_Id"])
        e = rows.setdefault(key, {"kernel": r["Kernel_Name"], "c": {}})
        e["c"][r["Counter_Name"]] = e["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return rows


def short(name):
    for k in ("k_extend<false, true", "k_extend<false, false", "k_shade<true>", "k_shade<false>", "k_tail", "k_resolve"):
        if k in name:
            return {"k_extend<false, true": "k_extend", "k_extend<false, false": "k_extend", "k_shade<true>": "k_shade",
                    "k_shade<false>": "k_shade"}.get(k, k)
    return None


def frame_dispatches(rows, levels):
    """[(kernel, level, counters)] of the render kernels of the frame: per batch and level k the launches are
    k_tail (k >= 1) | k_extend | k_shade; the counters of the batches of a frame are summed per level."""
    agg, seen = collections.OrderedDict(), {"k_extend": 0, "k_shade": 0, "k_tail": 0}
    for _, e in sorted(rows.items()):
        k = short(e["kernel"])
        if k in seen:
            per = levels if k != "k_tail" else levels - 1
            lv = seen[k] % per + (1 if k == "k_tail" else 0)
            seen[k] += 1
            c = agg.setdefault((k, lv), {})
            for name, v in e["c"].items():
                c[name] = c.get(name, 0.0) + v
    return [(k, lv, c) for (k, lv), c in agg.items()], seen["k_extend"] // max(1, levels)


def trace_durations(d, levels, batches):
    """Mean duration (ms) per (kernel, level) over the frames of the bench run, and the frame count."""
    fs = glob.glob(f"{d}/*/*kernel_trace.csv") + glob.glob(f"{d}/*kernel_trace.csv")
    seq = collections.defaultdict(list)
    for r in sorted(csv.DictReader(open(fs[0])), key=lambda r: int(r["Start_Timestamp"])):
        k = short(r["Kernel_Name"])
        if k in ("k_extend", "k_shade", "k_tail") and "k_extend<true" not in r["Kernel_Name"]:
            seq[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    dur = {}
    frames = len(seq["k_extend"]) // (levels * batches)
    for k, v in seq.items():
        per = levels if k != "k_tail" else levels - 1
        n = len(v) // (per * batches)
        for lv in range(per):
            # per frame: the sum over its batches; over the frames: the median (the count_work frame's k_shade launches differ)
            xs = [sum(v[(f * batches + b) * per + lv] for b in range(batches)) for f in range(n)]
            dur[(k, lv if k != "k_tail" else lv + 1)] = sorted(xs)[len(xs) // 2] if xs else 0.0
    return dur, frames


def main():
    raw, tag, workload = sys.argv[1:4]
    variant = sys.argv[4] if len(sys.argv) > 4 else None
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import bench
    levels = bench.WORKLOADS[workload][3] + 2        # rays per path <= depth + 2: launches per batch and kernel
    passes, batches = {}, 1
    for n in ("fetch", "write", "sq", "sq2", "tcc", "tcp", "grbm", "td"):
        passes[n], nb = frame_dispatches(counter_rows(os.path.join(raw, n)), levels)
        batches = max(batches, nb)
    dur, frames = trace_durations(os.path.join(raw, "stats"), levels, batches)
    merged = collections.OrderedDict()
    for n, disp in passes.items():
        for k, lv, c in disp:
            merged.setdefault((k, lv), {}).update(c)
    recs = []
    for (k, lv), c in merged.items():
        d_ms = dur.get((k, lv), 0.0)
        r = {"kernel": k, "level": lv, "duration_ms": d_ms, "counters": c}
        act, thr, busy = c.get("SQ_ACTIVE_INST_VALU", 0.0), c.get("SQ_THREAD_CYCLES_VALU", 0.0), c.get("SQ_BUSY_CYCLES", 0.0)
        if act > 1e5 and busy > 0:
            simd_cycles = busy / N_SE * N_SIMD
            r["valu_inst_rate"] = 4.0 * act / simd_cycles
            r["lanes_active"] = thr / act
            r["useful_lane_slots_4cyc"] = r["valu_inst_rate"] * r["lanes_active"] / 64.0
            r["lane_ops"] = thr
            r["cycles_per_valu_inst_per_simd"] = simd_cycles / max(1.0, c.get("SQ_INSTS_VALU", 0.0))
        if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
            r["hbm_bytes"] = (2.0 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024.0
        if c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0) > 0:
            r["l2_hit"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
        if c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) > 0:
            r["l1_hit"] = 1.0 - c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"]
            r["l2_read_bytes"] = c["TCP_TCC_READ_REQ_sum"] * 64.0
        if "GRBM_GUI_ACTIVE" in c and d_ms > 0:
            r["eff_clock_ghz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / (d_ms * 1e-3) / 1e9
        if "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"] > 0:   # share of the launch the 256 texture-data / texture-address units were busy
            if "TD_TD_BUSY_sum" in c:
                r["td_busy"] = c["TD_TD_BUSY_sum"] / 256.0 / (c["GRBM_GUI_ACTIVE"] / 8.0)
            if "TA_TA_BUSY_sum" in c:
                r["ta_busy"] = c["TA_TA_BUSY_sum"] / 256.0 / (c["GRBM_GUI_ACTIVE"] / 8.0)
        if "SQ_INSTS_FLAT" in c:
            r["flat_insts"] = c["SQ_INSTS_FLAT"]      # wave-level generic loads/stores (node and primitive fetches, hit stores)
        if "SQ_INSTS_VALU" in c:
            r["valu_insts"] = c["SQ_INSTS_VALU"]
        recs.append(r)

    def summary(kernel):
        rs = [r for r in recs if r["kernel"] == kernel and "lane_ops" in r]
        t = sum(r["duration_ms"] for r in rs)
        if not rs or t <= 0:
            return None
        wavg = lambda key: sum(r.get(key, 0.0) * r["duration_ms"] for r in rs) / t
        return {"launches_with_work": len(rs), "duration_ms_sum": t, "lane_ops_sum": sum(r["lane_ops"] for r in rs),
                "valu_inst_rate": wavg("valu_inst_rate"), "lanes_active": wavg("lanes_active"),
                "useful_lane_slots_4cyc": wavg("useful_lane_slots_4cyc"),
                "hbm_bytes_sum": sum(r.get("hbm_bytes", 0.0) for r in rs), "l2_read_bytes_sum": sum(r.get("l2_read_bytes", 0.0) for r in rs),
                "l2_hit": wavg("l2_hit"), "l1_hit": wavg("l1_hit"), "eff_clock_ghz": wavg("eff_clock_ghz"),
                "by_level": {str(r["level"]): {k: r[k] for k in ("duration_ms", "valu_inst_rate", "lanes_active", "useful_lane_slots_4cyc", "lane_ops",
                                                                     "hbm_bytes", "l2_read_bytes", "l2_hit", "l1_hit", "eff_clock_ghz",
                                                                     "cycles_per_valu_inst_per_simd", "td_busy", "ta_busy", "flat_insts", "valu_insts") if k in r} for r in rs}}

    n_ext = sum(1 for r in recs if r["kernel"] == "k_extend")
    n_ext *= batches
    rays_frame = None
    try:  # one_frame.py prints: workload ms_total ms_extend rays_cast
        rays_frame = int(open(os.path.join(raw, "sq.out")).read().split()[-1])
    except (OSError, ValueError, IndexError):
        pass
    out = {"tag": tag, "workload": workload, "variant": variant, "frames_in_trace": frames, "batches_per_frame": batches, "extend_launches_per_frame": n_ext,
           "definitions": __doc__.split("Every figure")[1].strip(),
           "summary": {k: summary(k) for k in ("k_extend", "k_shade", "k_tail")}, "dispatches": recs}
    with open(os.path.join("profiles", f"{tag}_pmc.json"), "w") as f:
        json.dump(out, f, indent=1)
    path = os.path.join("profiles", "roofline_pmc.json" if not variant else f"{tag}_roofline_variant.json")
    allw = json.load(open(path)) if os.path.exists(path) and not variant else {}
    # what ties the counters to the code they were taken from (bench.pmc_record refuses the record on a mismatch)
    abi = None
    try:
        from pathtracer_amd import _native
        abi = int(_native.mi.mi_abi_version())
    except Exception:  # noqa: BLE001  (converted on a machine without the library: the source hash still binds)
        pass
    allw[workload] = {"tag": tag, "source": f"profiles/{tag}_pmc.json", "kernel_source_sha256": bench.kernel_source_sha256(), "abi_version": abi,
                      "extend_launches_per_frame": n_ext, "rays_cast_frame": rays_frame, **{k: v for k, v in out["summary"].items()}}
    with open(path, "w") as f:
        json.dump(allw, f, indent=1)
    s = out["summary"]["k_extend"]
    print(json.dumps({k: s[k] for k in s if k != "by_level"}))
    for lv, r in s["by_level"].items():
        print("k_extend level", lv, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()})
    for kk in ("k_shade", "k_tail"):
        if out["summary"][kk]:
            print(kk, {k: v for k, v in out["summary"][kk].items() if k != "by_level"})


if __name__ == "__main__":
    main()
