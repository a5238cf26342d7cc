#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter values per kernel name from a counter_collection csv dir."""
import csv, glob, os, sys, collections
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            agg[r["Kernel_Name"].split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    if "skh" not in k:
        continue
    print(k, "launches=%d" % max(len(v) for v in cs.values()))
    for c, v in sorted(cs.items()):
        print("   %-28s %16.0f" % (c, sum(v) / len(v)))
