"""Turns the FETCH_SIZE / WRITE_SIZE PMC passes of `bench.py --steps 1 --warmup 0`
into profiles/traffic.json (HBM bytes per k_extend launch).
usage: python scratch/traffic.py <fetch_dir> <write_dir> <workload> <steps_profiled>"""
import csv, glob, json, os, sys
fetch_dir, write_dir, workload = sys.argv[1:4]
def kb(d, counter):
    f = (glob.glob(f"{d}/*/*counter_collection.csv") + glob.glob(f"{d}/*counter_collection.csv"))[0]
    tot, n = 0.0, 0
    for r in csv.DictReader(open(f)):
        if "k_extend<false" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            tot += float(r["Counter_Value"]); n += 1
    return tot, n
fk, n1 = kb(fetch_dir, "FETCH_SIZE")
wk, n2 = kb(write_dir, "WRITE_SIZE")
assert n1 == n2 and n1 > 0
per_launch = (2.0 * fk + wk) * 1024.0 / n1     # FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE reads half on gfx950
path = "profiles/traffic.json"
rec = json.load(open(path)) if os.path.exists(path) else {}
rec[workload] = {"hbm_bytes_per_launch": per_launch, "launches": n1, "fetch_kib_raw": fk, "write_kib": wk,
                 "note": "FETCH_SIZE doubled (MI355X_MICROARCH.md HBM section), separate --pmc passes"}
json.dump(rec, open(path, "w"), indent=1)
print(rec[workload])
