"""summarise rocprofv3 counter_collection.csv per kernel (dev tool): python scripts/dev/pmc_summary.py dir..."""
import csv, sys, collections, glob
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*counter_collection.csv"):
        rows = list(csv.DictReader(open(f)))
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in rows:
            acc[r["Kernel_Name"][:28]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            if "brov" not in k: continue
            print(k, {c: round(sum(v[5:]) / max(1, len(v[5:]))) for c, v in cs.items()}, "launches", len(next(iter(cs.values()))))
