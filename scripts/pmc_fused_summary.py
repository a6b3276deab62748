import csv, glob, re, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(trace_kernel|bounce_kernel|fused_kernel)", r["Kernel_Name"])
        if m:
            agg[m[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    print(k, {n: round(sum(v[-10:]) / len(v[-10:])) for n, v in c.items()}, "launches", len(next(iter(c.values()))))
