import csv,glob,collections,sys
d=sys.argv[1]
f=glob.glob(f"{d}/*/*counter_collection.csv")[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0][-28:]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
for k,v in agg.items():
    if k.startswith("__amd") or "elementwise" in k or "sum_counts" in k: continue
    print(k, {c:f"{x:.4g}" for c,x in v.items()}, "n=",max(cnt[(k,c)] for c in v))
