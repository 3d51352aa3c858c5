#!/usr/bin/env python3
"""Turn a rocprofv3 --kernel-trace --stats kernel_stats.csv into a short table (kernel names shortened).

usage: tools/summarize_profile.py gpurun_out/prof/vm/*_kernel_stats.csv "command line" > profiles/rNN_name.md
"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"rocprim::ROCPRIM_\d+_NS::", "rocprim::", name)
    m = re.search(r"rocprim::detail::(\w+)", name)
    if "rocprim" in name and m:
        inner = re.findall(r"rocprim::detail::(\w+_impl|radix_sort_\w+|scan_impl|init_lookback_scan_state_kernel|\w+_kernel)", name)
        return "rocprim::" + (inner[1] if len(inner) > 1 else inner[0] if inner else m.group(1))
    name = re.sub(r"\(.*", "", name)
    return name.replace("void ", "")


def load(path):
    """kernel_stats.csv rows, or the same aggregate computed from a rocpd results .db (rocprofv3's default output)."""
    if not path.endswith(".db"):
        return list(csv.DictReader(open(path)))
    import sqlite3
    con = sqlite3.connect(path)
    per = {}
    for name, dur in con.execute("select name, duration from kernels"):
        a = per.setdefault(name, [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
    total = sum(a[1] for a in per.values()) or 1
    return [{"Name": k, "Calls": a[0], "TotalDurationNs": a[1], "MinNs": a[2], "MaxNs": a[3], "Percentage": 100.0 * a[1] / total}
            for k, a in per.items()]


rows = load(sys.argv[1])
print(f"# rocprofv3 --kernel-trace --stats\n\ncommand: `{sys.argv[2] if len(sys.argv) > 2 else ''}`\n")
print("| kernel | calls | total ms | avg us | % | min us | max us |")
print("|---|---:|---:|---:|---:|---:|---:|")
agg = {}
for r in rows:
    k = short(r["Name"])
    a = agg.setdefault(k, [0, 0.0, 1e30, 0.0, 0.0])
    a[0] += int(r["Calls"]); a[1] += float(r["TotalDurationNs"]); a[2] = min(a[2], float(r["MinNs"])); a[3] = max(a[3], float(r["MaxNs"]))
    a[4] += float(r["Percentage"])
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"| `{k}` | {a[0]} | {a[1] / 1e6:.3f} | {a[1] / a[0] / 1e3:.1f} | {a[4]:.2f} | {a[2] / 1e3:.1f} | {a[3] / 1e3:.1f} |")
