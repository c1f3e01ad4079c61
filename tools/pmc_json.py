"""Per-kernel averages of rocprofv3 --pmc passes (one directory per pass) as JSON:
   python tools/pmc_json.py <out.json> <dir> [<dir> ...]
Only the largest-grid dispatches of each kernel are averaged (the solve kernels also run as tiny activation passes)."""
import collections, csv, glob, json, re, sys

out, dirs = sys.argv[1], sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        big = collections.defaultdict(int)
        for r in rows:
            big[r["Kernel_Name"]] = max(big[r["Kernel_Name"]], int(r["Grid_Size"]))
        for r in rows:
            k = r["Kernel_Name"]
            if "daqp_amd" not in k or int(r["Grid_Size"]) < big[k]:
                continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, cs in acc.items():
    name = re.sub(r"^void daqp_amd::|\(.*$", "", k)
    # (the solve kernels are also launched as activation passes that return at once, and a warm-started sequence starts with one
    #  cold solve: drop the near-empty dispatches of each counter, then keep those around the median of the rest)
    def typical(v):
        w = sorted(x for x in v if x >= 0.02 * max(v)) if max(v) > 0 else list(v)
        med = w[len(w) // 2]
        return [x for x in w if 0.5 * med <= x <= 1.5 * med] or w
    heavy = {c: typical(v) for c, v in cs.items()}
    res[name] = {c: {"mean": sum(v) / len(v), "dispatches": len(v)} for c, v in sorted(heavy.items())}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: {c: round(v["mean"]) for c, v in cs.items()} for k, cs in res.items()}, indent=1))
