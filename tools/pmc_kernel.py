"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel name (developer tool)."""
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:48]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        n[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {v / n[(k, c)]:16.0f} /launch")
