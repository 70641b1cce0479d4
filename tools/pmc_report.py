"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel (mean per dispatch)."""
import collections, csv, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
flt = sys.argv[2:] or [""]
for k in sorted(agg):
    if any(f in k for f in flt):
        print(k, {c: round(v / cnt[(k, c)], 1) for c, v in agg[k].items()}, "dispatches:", max(cnt[(k, c)] for c in agg[k]))
