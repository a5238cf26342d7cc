#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter values per kernel name from a counter_collection csv dir."""
import csv, glob, os, re, sys, collections
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            m = re.search(r"(k_\w+)(<[^>]*>)?", r["Kernel_Name"])        # (kernels of an unnamed namespace have a "(" in their names)
            name = "skh::" + m.group(0) if m and "skh" in r["Kernel_Name"] else r["Kernel_Name"].split("(")[0][-60:]
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    if "skh" not in k:
        continue
    print(k, "launches=%d" % max(len(v) for v in cs.values()))
    for c, v in sorted(cs.items()):
        print("   %-28s %16.0f" % (c, sum(v) / len(v)))
