#!/usr/bin/env python3
"""Builds profiles/<tag>_hbm_traffic.json from two rocprofv3 --pmc passes over the same bench
command (one with TCC_EA0_RDREQ-derived FETCH_SIZE, one with WRITE_SIZE): per-kernel averages per
launch.  usage: make_traffic_json.py <fetch_dir> <write_dir> <out.json> <nrec> <read_len>"""
import collections, csv, glob, hashlib, json, os, sys

def averages(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == counter and "skh::" in r["Kernel_Name"]:
                    agg[r["Kernel_Name"].split("(")[0].replace("void ", "").strip()].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}

fetch_dir, write_dir, out, nrec, L = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
fetch, write = averages(fetch_dir, "FETCH_SIZE"), averages(write_dir, "WRITE_SIZE")
total = nrec * L
algorithmic = {"k_reverse_staged": 2 * total + 8 * (nrec + 1), "k_reverse_short": 2 * total + 8 * (nrec + 1),
               "k_count_short": total + 8 * (nrec + 1) + 32 * nrec, "k_stats": total + 8 * (nrec + 1)}
kernels = {}
for k in sorted(set(fetch) & set(write)):
    if fetch[k] < 1024:          # the early-exit tile kernels
        continue
    alg = next((v for n, v in algorithmic.items() if n in k), None)
    kernels[k] = {"FETCH_SIZE_KB": round(fetch[k]), "WRITE_SIZE_KB": round(write[k]),
                  "hbm_bytes": int(2 * fetch[k] * 1024 + write[k] * 1024), "algorithmic_bytes": alg}
doc = {"note": "rocprofv3 --pmc, two separate passes (FETCH_SIZE, then WRITE_SIZE) over `python3 bench.py --steps 3 "
               "--warmup 1` on MI355X, kernel filter skh; per-launch averages in KB as reported.  gfx950 correction: "
               "FETCH_SIZE counts 128-B read requests as 64 B for wide coalesced streams, so bytes_read = 2 * "
               "FETCH_SIZE * 1024; WRITE_SIZE * 1024 is exact.",
       "workload": "%d x %d bp, one plane per launch" % (nrec, L),
       # bench.py reports these figures only while kernels.hip is the file they were measured on
       "kernels_sha16": hashlib.sha256(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "seqkit_amd", "csrc",
                                                         "kernels.hip"), "rb").read()).hexdigest()[:16],
       "kernels": kernels}
with open(out, "w") as fh:
    json.dump(doc, fh, indent=1)
print(json.dumps(doc, indent=1))
