#!/usr/bin/env python3
"""Per-bounce extend/shade durations of the last step from a rocprofv3 kernel_trace.csv."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for key in ('k_wf_extend<false>', 'k_wf_shade', 'k_wf_generate', 'k_megakernel<0, false>'):
    k = [r for r in rows if key in r['Kernel_Name']]
    if k:
        print(key, [round(dur(r)) for r in k[-10:]], 'sum', round(sum(dur(r) for r in k[-10:])))
