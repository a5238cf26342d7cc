#!/usr/bin/env python3
"""Condenses a rocprofv3 --kernel-trace --stats output directory into a small text table
(per kernel: calls, total/avg/min/max duration) that is committed under profiles/."""
import csv, glob, os, sys
d = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
    with open(f) as fh:
        rows += list(csv.DictReader(fh))
if not rows:
    # fall back to the raw trace
    agg = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                a = agg.setdefault(r["Kernel_Name"], [0, 0, 1 << 62, 0])
                a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
    rows = [{"Name": k, "Calls": v[0], "TotalDurationNs": v[1], "AverageNs": v[1] / v[0], "MinNs": v[2], "MaxNs": v[3]}
            for k, v in agg.items()]
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print("%-70s %6s %14s %12s %12s %12s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns"))
for r in rows:
    print("%-70s %6s %14s %12.0f %12s %12s" % (r["Name"][:70], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]),
                                               r["MinNs"], r["MaxNs"]))
