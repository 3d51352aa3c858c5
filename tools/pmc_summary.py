#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection.csv files per kernel (sum over dispatches / number of dispatches).
usage: tools/pmc_summary.py dir1 [dir2 ...]   (each dir = one --pmc pass)"""
import collections, csv, glob, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
            if "rocprim" in k or "rocclr" in k: continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        n = len(calls[(k, c)])
        print(f"    {c:32s} {agg[k][c] / n:18.1f} per dispatch  ({n} dispatches)")
