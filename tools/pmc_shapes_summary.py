#!/usr/bin/env python3
"""Condenses the passes of tools/pmc_shapes.sh into gpurun_out/<tag>_hbm_traffic_shapes.json (HBM bytes per
launch against the algorithmic bytes, per shape and kernel) and gpurun_out/<tag>_sq_counters.txt."""
import collections, csv, glob, json, os, sys
import numpy as np

tag = sys.argv[1]
GRCH38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717,
          133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285,
          58617616, 64444167, 46709983, 50818468, 156040895, 57227415]


def averages(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == counter and "skh::" in r["Kernel_Name"]:
                    agg[r["Kernel_Name"].split("(")[0].replace("void ", "").strip()].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def shape(which):
    if which == "chrom":
        return len(GRCH38), sum(GRCH38)
    if which == "ont":
        rng = np.random.default_rng(0xC4)
        lens = np.clip(np.exp(rng.normal(np.log(8000), 1.1, 700_000)), 1000, 1_000_000).astype(np.int64)
        return len(lens), int(lens.sum())
    rng = np.random.default_rng(11)
    lens = np.full(4_000_000, 150); lens[rng.integers(0, len(lens), 200)] = rng.integers(100_000, 3_000_000, 200)
    return len(lens), int(lens.sum())


doc = {"note": "rocprofv3 --pmc, separate passes for FETCH_SIZE and WRITE_SIZE over tools/microbench_long.py (WHICH = shape) on "
               "MI355X; per-launch averages.  gfx950: bytes read = 2 * FETCH_SIZE * 1024 (128-byte requests are tallied at "
               "64), bytes written = WRITE_SIZE * 1024.  A launch of the library is several kernels (prepare, short records, "
               "long records): their traffic is listed per kernel, `sum_hbm_bytes` adds the kernels of one operation.",
       "shapes": {}}
for which in ("chrom", "ont", "mixed"):
    nrec, total = shape(which)
    fetch = averages("gpurun_out/pmc_%s_FETCH_SIZE" % which, "FETCH_SIZE")
    write = averages("gpurun_out/pmc_%s_WRITE_SIZE" % which, "WRITE_SIZE")
    ks = {k: {"hbm_bytes": int(2 * fetch[k] * 1024 + write.get(k, 0) * 1024), "FETCH_SIZE_KB": round(fetch[k]), "WRITE_SIZE_KB": round(write.get(k, 0))}
          for k in sorted(fetch)}
    ops = {"revcomp": (2 * total + 8 * nrec, ("k_reverse", "k_prepare")), "count4": (total + 40 * nrec, ("k_count", "k_prepare")),
           "stats": (total + 8 * nrec, ("k_stats",))}
    doc["shapes"][which] = {"records": nrec, "bases": total, "kernels": ks,
                            "operations": {op: {"algorithmic_bytes": alg,
                                                "sum_hbm_bytes": sum(v["hbm_bytes"] for k, v in ks.items() if any(p in k for p in pre) and
                                                                     not (op != "revcomp" and "k_prepare" in k and False))}
                                           for op, (alg, pre) in ops.items()}}
    # k_prepare runs once per revcomp and once per count launch: negligible (a few KiB); it is counted in both sums
    for op, o in doc["shapes"][which]["operations"].items():
        o["ratio"] = round(o["sum_hbm_bytes"] / o["algorithmic_bytes"], 3)
with open("gpurun_out/%s_hbm_traffic_shapes.json" % tag, "w") as fh:
    json.dump(doc, fh, indent=1)
print(json.dumps({w: s["operations"] for w, s in doc["shapes"].items()}, indent=1))

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_sq/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if "skh::" in r["Kernel_Name"]:
                agg[r["Kernel_Name"].split("(")[0].replace("void ", "").strip()][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/%s_sq_counters.txt" % tag, "w") as fh:
    fh.write("# rocprofv3 --pmc over `bench.py --steps 3 --warmup 1` (10 M x 150 bp), per-launch averages; SQ_* in quad-cycles summed over waves\n")
    for k, cs in sorted(agg.items()):
        wc = sum(cs.get("SQ_WAVE_CYCLES", [0])) / max(len(cs.get("SQ_WAVE_CYCLES", [1])), 1)
        fh.write("%s  launches=%d\n" % (k, max(len(v) for v in cs.values())))
        for c, v in sorted(cs.items()):
            a = sum(v) / len(v)
            fh.write("   %-24s %16.0f  %6.1f%% of wave cycles\n" % (c, a, 100 * a / wc if wc else 0))
print(open("gpurun_out/%s_sq_counters.txt" % tag).read())
