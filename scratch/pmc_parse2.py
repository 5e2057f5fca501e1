import csv,glob,collections,sys
d=sys.argv[1]
f=(glob.glob(f"{d}/*/*counter_collection.csv")+glob.glob(f"{d}/*counter_collection.csv"))[0]
rows=collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    key=(int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0][-40:])
    rows[key][r["Counter_Name"]]=rows[key].get(r["Counter_Name"],0.0)+float(r["Counter_Value"])
for (i,k),v in sorted(rows.items()):
    if "k_extend" in k or "k_shade" in k or "k_tail" in k:
        if max(v.values())<1e5: continue
        print(i,k[-34:], {c:f"{x:.4g}" for c,x in v.items()})
