"""Summarise rocprofv3 --pmc CSV output per kernel: python scripts/pmc_summary.py <dir> [kernel-substring]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "k_trace_shade"
tot = collections.defaultdict(float)
n = collections.defaultdict(int)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        if pat not in r["Kernel_Name"]:
            continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
        key = (f, r["Dispatch_Id"])
        if (key, r["Counter_Name"]) not in seen:
            seen.add((key, r["Counter_Name"]))
            n[r["Counter_Name"]] += 1
for k in sorted(tot):
    print(f"{k:28s} total {tot[k]:.6g}  dispatches {n[k]}  per-dispatch {tot[k] / max(n[k], 1):.6g}")
