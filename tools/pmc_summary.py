"""Average rocprofv3 PMC counters per dispatch for kernels matching a regex (largest grids only)."""
import csv, glob, collections, sys, re
d, pat = sys.argv[1], re.compile(sys.argv[2])
mingrid = int(sys.argv[3]) if len(sys.argv) > 3 else 0
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
if not files:
    print("no counter_collection.csv under", d, glob.glob(d + "/**/*", recursive=True)[:10])
for f in files:
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if pat.search(r["Kernel_Name"]) and int(r["Grid_Size"]) >= mingrid:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k in sorted(acc):
        print(f"{k:28s} {acc[k] / n[k]:16.1f}  (dispatches {n[k]})")
