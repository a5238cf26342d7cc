#!/usr/bin/env python3
"""profiles/<tag>_hbm_traffic_<group>.json from the passes of tools/pmc_groups.sh (hot-path kernels per length
distribution and operation) and tools/pmc_text.sh (text stages): per kernel the bytes per launch the counters saw
next to the algorithmic bytes of the operation, and the hash of the source file the kernel lives in.
  usage: make_traffic_groups.py <tag> <pmc_groups outdir> <pmc_text outdir>"""
import collections, csv, glob, hashlib, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "seqkit_amd", "csrc")
NOTE = ("rocprofv3 --pmc on MI355X, one counter per pass (FETCH_SIZE, then WRITE_SIZE), kernel filter skh; per-launch "
        "averages in KB as reported.  gfx950 correction: FETCH_SIZE tallies the 128-byte read requests of wide coalesced "
        "streams as 64 bytes, so hbm_bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 (exact for 16-byte-per-lane "
        "streams; narrower or scattered reads are not calibrated, their ratio is an upper bound).  One operation per "
        "pass: nothing is summed over operations.  \"operation\": the bytes of all kernels of the one operation (the "
        "short-record and the long-record kernel share the records of a batch) against the algorithmic bytes the "
        "operation has to move; ratio 1.0 = every byte crosses the memory interface once.")


def sha16(path):
    with open(path, "rb") as fh:
        return hashlib.sha256(fh.read()).hexdigest()[:16]


def source_of(kernel):
    base = re.search(r"k_\w+", kernel).group(0)
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".hip") and re.search(r"\b%s\b\s*\(" % base, open(os.path.join(CSRC, f)).read()):
            return f
    return None


def averages(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == counter and "skh" in r["Kernel_Name"]:
                    m = re.search(r"(k_\w+)(<[^>]*>)?", r["Kernel_Name"])
                    if m:
                        agg["skh::" + m.group(0)].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def kernels_of(fetch_dir, write_dir, algorithmic):
    fetch, write = averages(fetch_dir, "FETCH_SIZE"), averages(write_dir, "WRITE_SIZE")
    out, files = {}, {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, (0, 0))[0], write.get(k, (0, 0))[0]
        if 2 * f + w < 64:                                  # launches that leave at once
            continue
        src = source_of(k)
        files[src] = sha16(os.path.join(CSRC, src)) if src else None
        e = {"FETCH_SIZE_KB": round(f), "WRITE_SIZE_KB": round(w), "hbm_bytes": int((2 * f + w) * 1024),
             "launches": fetch.get(k, write.get(k))[1], "source": src}
        out[k] = e
    if algorithmic:
        # the operation as a whole: its kernels share the records between them (short records / long records), so
        # the bytes of all of them against the bytes the operation has to move
        tot = sum(e["hbm_bytes"] for e in out.values())
        for e in out.values():
            e["share_of_operation"] = round(e["hbm_bytes"] / tot, 3) if tot else 0
        out["operation"] = {"hbm_bytes": tot, "algorithmic_bytes": int(algorithmic), "ratio": round(tot / algorithmic, 3)}
    return out, files


def shape_header(log):
    for line in open(log):
        m = re.match(r"== (.*): (\d+) records, ([\d.]+) Gbases, max (\d+)", line)
        if m:
            return m.group(1), int(m.group(2)), float(m.group(3)), int(m.group(4))
    return None


def main():
    tag, gdir, tdir = sys.argv[1], sys.argv[2], sys.argv[3]
    groups = {"short_records": ["r100", "r150", "r250", "trim", "mirna"], "long_records": ["mixed", "chrom", "ont", "kbtail"]}
    for gname, shapes in groups.items():
        doc = {"note": NOTE, "shapes": {}, "sources": {}}
        for sh in shapes:
            for op in ("revcomp", "inplace", "count4", "stats"):
                fd, wd = os.path.join(gdir, "%s_%s_FETCH_SIZE" % (sh, op)), os.path.join(gdir, "%s_%s_WRITE_SIZE" % (sh, op))
                if not (os.path.isdir(fd) and os.path.isdir(wd)):
                    continue
                hdr = shape_header(fd + ".log")
                if not hdr:
                    continue
                name, nrec, gb, mx = hdr
                total = None
                # exact totals are not in the header line: take them from the timing line's GB/s * ms
                for line in open(fd + ".log"):
                    m = re.match(r"\s+%s\s+([\d.]+) ms\s+([\d.]+) GB/s" % op, line)
                    if m:
                        total = float(m.group(1)) * float(m.group(2)) * 1e6      # the operation's algorithmic bytes as the tool counts them
                ks, files = kernels_of(fd, wd, total)
                doc["sources"].update(files)
                doc["shapes"].setdefault(sh, {"distribution": name, "records": nrec, "gbases": gb, "longest": mx, "operations": {}})
                doc["shapes"][sh]["operations"][op] = ks
        doc["kernels_sha16"] = doc["sources"].get("kernels.hip")
        if doc["shapes"]:
            with open(os.path.join(ROOT, "profiles", "%s_hbm_traffic_%s.json" % (tag, gname)), "w") as fh:
                json.dump(doc, fh, indent=1)
    # text stages
    fd, wd = os.path.join(tdir, "pmc_FETCH_SIZE"), os.path.join(tdir, "pmc_WRITE_SIZE")
    if os.path.isdir(fd) and os.path.isdir(wd):
        ks, files = kernels_of(fd, wd, None)
        doc = {"note": NOTE + "  Text stages: tools/microbench_text.py, 4,000,000 records x 150 (FASTQ 1.36 GB and FASTA 0.75 GB of text; "
                              "kernels without a <true>/<false> suffix are averaged over both files).",
               "sources": files, "kernels": ks}
        with open(os.path.join(ROOT, "profiles", "%s_hbm_traffic_text.json" % tag), "w") as fh:
            json.dump(doc, fh, indent=1)


main()
