#!/usr/bin/env python3
"""Times individual entry points on a resident 10M x 150 batch (GPU only): isolates where a
kernel's time goes (window loads vs table lookups vs stores)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import seqkit_amd
from bench import synth_fastq

nrec = int(os.environ.get("NREC", 10_000_000)); L = int(os.environ.get("READLEN", 150))
dev = torch.device("cuda", 0)
lib = seqkit_amd.load(); ctx = lib.init(0)
seq, qual, offs = synth_fastq(nrec, L, 0xC2, dev)
total = nrec * L
so, qo = torch.empty_like(seq), torch.empty_like(qual)
counts = torch.empty(nrec * 8, dtype=torch.int64, device=dev)
stats = torch.zeros(5, dtype=torch.int64, device=dev)
lut = lib.pair_table(1); cls4 = lib.class_table(["G", "C", "A", "T"]); cls6 = lib.class_table(["G", "C", "A", "T", "N", "-"])
gap = (C.c_uint8 * 256)(); gap[ord("-")] = 1
valid = (C.c_uint8 * 256)(); lib.check(lib.skh_valid_table(1, valid))
sub_offs = torch.empty(nrec + 1, dtype=torch.int64, device=dev)
sub_wsb = lib.skh_subseq_workspace_size(nrec)
sub_ws_t = torch.empty(sub_wsb + 256, dtype=torch.uint8, device=dev); sub_ws = (sub_ws_t.data_ptr() + 255) & ~255
sub_res = torch.zeros(3, dtype=torch.int64, device=dev)


def subseq(a, b, minus):
    return lib.skh_subseq_dev(ctx, p(seq), None, p(offs), nrec, total, None, nrec, a, b, minus, lut, p(so), None, total,
                              p(sub_offs), C.c_void_p(sub_ws), sub_wsb, p(sub_res), sp)


gap_offs = torch.empty(nrec + 1, dtype=torch.int64, device=dev)
gap_wsb = lib.skh_remove_gaps_workspace_size(nrec, total)
gap_ws_t = torch.empty(gap_wsb + 256, dtype=torch.uint8, device=dev); gap_ws = (gap_ws_t.data_ptr() + 255) & ~255
gap_n = (C.c_uint8 * 256)(); gap_n[ord("N")] = 1; gap_n[ord("-")] = 1


def remove_gaps(with_qual):
    return lib.skh_remove_gaps_dev(ctx, p(seq), p(qual) if with_qual else None, p(offs), nrec, total, gap_n, p(so),
                                   p(qo) if with_qual else None, p(gap_offs), C.c_void_p(gap_ws), gap_wsb, p(sub_res), sp)


torch.cuda.synchronize()   # data was generated on the default stream
st = torch.cuda.Stream(device=dev); sp = C.c_void_p(st.cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
GB = 1e9
cases = {
    "revcomp_fastq (4B/base)": (lambda: lib.skh_revcomp_fastq_dev(ctx, p(seq), p(so), p(qual), p(qo), p(offs), nrec, total, lut, 1, sp), 4 * total),
    "revcomp seq only (2B/base)": (lambda: lib.skh_revcomp_dev(ctx, p(seq), p(so), p(offs), nrec, total, lut, 1, sp), 2 * total),
    "reverse only, no table (2B/base)": (lambda: lib.skh_revcomp_dev(ctx, p(seq), p(so), p(offs), nrec, total, None, 1, sp), 2 * total),
    "reverse fastq, no table (4B/base)": (lambda: lib.skh_revcomp_fastq_dev(ctx, p(seq), p(so), p(qual), p(qo), p(offs), nrec, total, None, 1, sp), 4 * total),
    "complement only (aligned, 2B/base)": (lambda: lib.skh_revcomp_dev(ctx, p(seq), p(so), p(offs), nrec, total, lut, 0, sp), 2 * total),
    "count 4 classes (1B/base+40B/rec)": (lambda: lib.skh_count_dev(ctx, p(seq), p(offs), nrec, total, cls4, 4, p(counts), sp), total + 40 * nrec),
    "count 6 classes (1B/base+56B/rec)": (lambda: lib.skh_count_dev(ctx, p(seq), p(offs), nrec, total, cls6, 6, p(counts), sp), total + 56 * nrec),
    "validate DNAredundant, all bytes (1B/base)": (lambda: lib.skh_validate_dev(ctx, p(seq), p(offs), nrec, total, valid, 0, p(stats), sp), total),
    "subseq -r 13:-13 (2B/out base+16B/rec)": (lambda: subseq(13, -13, 0), 2 * (L - 24) * nrec + 16 * nrec),
    "subseq -r 1:-1 minus strand (2B/base+16B/rec)": (lambda: subseq(1, -1, 1), 2 * total + 16 * nrec),
    "subseq -r 1:12 (2B/out base+16B/rec)": (lambda: subseq(1, 12, 0), 2 * 12 * nrec + 16 * nrec),
    "remove gaps, bases only (3B/base+24B/rec)": (lambda: remove_gaps(False), 3 * total + 24 * nrec),
    "remove gaps, bases + qualities (6B/base+24B/rec)": (lambda: remove_gaps(True), 6 * total + 24 * nrec),
    "stats (1B/base+8B/rec)": (lambda: lib.skh_stats_dev(ctx, p(seq), p(offs), nrec, total, gap, p(stats), sp), total + 8 * nrec),
}
with torch.cuda.stream(st):
    cases["torch copy_ (2B/base)"] = (lambda: (so.copy_(seq), 0)[1], 2 * total)
    only = os.environ.get("CASES")          # comma-separated substrings of case names
    for name, (fn, nbytes) in cases.items():
        if only and not any(w in name for w in only.split(",")):
            continue
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(10):
            fn()
        e1.record(st)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("%-40s %8.3f ms  %7.1f GB/s  %5.1f%% of 8 TB/s  %8.1f Gbases/s" % (name, ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / 80, total / ms / 1e6))
lib.skh_destroy(ctx)
