"""Sum PMC counters of a rocprofv3 --pmc run per kernel (counter_collection.csv): python tools/pmc_kernel.py <dir> [substr]"""
import csv, sys, glob, collections
d, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if sub in k:
            agg[k[:70]][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k[:70], r["Counter_Name"])] += 1
    for k, v in agg.items():
        print(k)
        for c, x in sorted(v.items()):
            print(f"   {c:32s} {x / cnt[(k, c)]:16.1f} per launch ({cnt[(k, c)]} launches)")
