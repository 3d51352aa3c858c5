#!/usr/bin/env python3
"""Timeline of the LAST build in a rocprofv3 --kernel-trace CSV of tools/build_only.py: start offset, duration and the idle
gap before every kernel.  usage: tools/build_timeline.py <..._kernel_trace.csv> [first kernel name]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
first = sys.argv[2] if len(sys.argv) > 2 else "k_init_bounds"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
seg = rows[starts[-1]:]
t0 = int(seg[0]["Start_Timestamp"]); prev_end = t0; busy = 0
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-60:]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  {name}")
    prev_end = max(prev_end, e); busy += e - s
print(f"total {(prev_end - t0) / 1e3:.1f} us, kernels {busy / 1e3:.1f} us, {len(seg)} launches")
