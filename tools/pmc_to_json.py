#!/usr/bin/env python3
"""rocprofv3 --pmc passes (tools/pmc_passes.sh) -> one JSON: per kernel, counters averaged per dispatch.

usage: tools/pmc_to_json.py gpurun_out/pmcN > profiles/rNN_pmc.json
HBM traffic per launch follows MI355X_MICROARCH.md (HBM / rocprofv3): FETCH_SIZE and WRITE_SIZE are KiB from separate
passes; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so fetch bytes = 2 * FETCH_SIZE * 1024."""
import collections, csv, glob, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from computeraytracer_amd import abi      # code_rev(): which kernel sources these counters belong to
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        if "rocprim" in k or "rocclr" in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
out = {}
for k in sorted(agg):
    c = {n: agg[k][n] / len(calls[(k, n)]) for n in sorted(agg[k])}
    d = {"dispatches": max(len(calls[(k, n)]) for n in agg[k]), "counters_per_dispatch": {n: round(v, 1) for n, v in c.items()}}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        d["hbm_traffic_bytes_per_dispatch"] = round(2 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024)
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
        d["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
    out[k] = d
json.dump({"source": sys.argv[1], "code_rev": abi.code_rev(), "kernels": out}, sys.stdout, indent=1)
