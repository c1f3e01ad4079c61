"""Per-dispatch rocprofv3 PMC values for kernels matching a regex: python tools/pmc_dispatches.py <dir> <regex>"""
import csv, glob, sys, re
d, pat = sys.argv[1], re.compile(sys.argv[2])
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat.search(r["Kernel_Name"]):
            print(r["Dispatch_Id"], r["Kernel_Name"][:60], r["Grid_Size"], r["Counter_Name"], r["Counter_Value"])
