#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --pmc pass with
TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TCP_TOTAL_ACCESSES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCP_PENDING_STALL_CYCLES GRBM_GUI_ACTIVE
(the vector-memory path: texture addresser, L1).  usage: texture_counters_summary.py <title>=<dir> ..."""
import collections, csv, glob, re, sys

print("# rocprofv3 --pmc TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TCP_TOTAL_ACCESSES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ "
      "TCP_PENDING_STALL_CYCLES GRBM_GUI_ACTIVE")
print("# per-launch averages, summed over the chip's units (GRBM_GUI_ACTIVE over 8 XCDs, TA / TCP over 256 CUs)")
for arg in sys.argv[1:]:
    title, d = arg.split("=", 1)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(k_\w+(?:<[^>]*>)?)", r["Kernel_Name"])
            agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("== " + title)
    for k, cs in sorted(agg.items(), key=lambda kv: -sum(kv[1]["GRBM_GUI_ACTIVE"]) / len(kv[1]["GRBM_GUI_ACTIVE"])):
        a = {c: sum(v) / len(v) for c, v in cs.items()}
        if a["GRBM_GUI_ACTIVE"] < 1e6:
            continue
        cyc = a["GRBM_GUI_ACTIVE"] / 8
        print("%-32s cycles %9.0f  TA busy %4.0f%%  TCP accesses %10.3g (%4.2f per TCP and cycle)  TCC read req %9.3g  "
              "write req %9.3g  TCP pending-stall %4.0f%%" % (
                  k, cyc, 100 * a["TA_TA_BUSY"] / (256 * cyc), a["TCP_TOTAL_ACCESSES"], a["TCP_TOTAL_ACCESSES"] / (256 * cyc),
                  a["TCP_TCC_READ_REQ"], a["TCP_TCC_WRITE_REQ"], 100 * a["TCP_PENDING_STALL_CYCLES"] / (256 * cyc)))
