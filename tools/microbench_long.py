#!/usr/bin/env python3
"""Long-record workloads on a resident batch (GPU only): chromosome-scale FASTA (config 5)
and ONT-like skewed read lengths (config 4).  Times the device entry points and checks
size-independent properties (revcomp twice = identity; counts add up; stats totals)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import seqkit_amd

dev = torch.device("cuda", 0)
lib = seqkit_amd.load(); ctx = lib.init(0)
p = lambda t: C.c_void_p(t.data_ptr())

GRCH38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717,
          133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285,
          58617616, 64444167, 46709983, 50818468, 156040895, 57227415]


def synth_plane(total, seed, soft=True):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    padded = (total + 15) & ~15
    seq = torch.zeros(padded, dtype=torch.uint8, device=dev)
    letters = torch.tensor(list(b"ACGTacgtNn-."), dtype=torch.uint8, device=dev)
    w = torch.tensor([.24, .24, .24, .24, .0075, .0075, .0075, .0075, .004, .004, .001, .001], device=dev)
    step = 1 << 26
    for a in range(0, total, step):
        b = min(total, a + step)
        idx = torch.multinomial(w, b - a, replacement=True, generator=g)
        seq[a:b] = letters[idx]
    return seq


def run(name, lengths, seed):
    lengths = np.asarray(lengths, np.int64)
    offs_h = np.zeros(len(lengths) + 1, np.int64); np.cumsum(lengths, out=offs_h[1:])
    total, nrec = int(offs_h[-1]), len(lengths)
    seq = synth_plane(total, seed)
    offs = torch.from_numpy(offs_h).to(dev)
    out, back = torch.empty_like(seq), torch.empty_like(seq)
    work = seq.clone()
    counts = torch.empty(nrec * 4, dtype=torch.int64, device=dev)
    stats = torch.zeros(5, dtype=torch.int64, device=dev)
    lut = lib.pair_table(1); cls = lib.class_table(["G", "C", "A", "T"])
    gap = (C.c_uint8 * 256)()
    for ch in b"- .": gap[ch] = 1
    torch.cuda.synchronize()
    st = torch.cuda.Stream(device=dev); sp = C.c_void_p(st.cuda_stream)
    ops = {
        "revcomp": (lambda: lib.skh_revcomp_dev(ctx, p(seq), p(out), p(offs), nrec, total, lut, 1, sp), 2 * total + 8 * nrec),
        # in place: `work` holds the batch, reversed once more by every call (same work each time)
        "inplace": (lambda: lib.skh_revcomp_inplace_dev(ctx, p(work), None, p(offs), nrec, total, lut, 1, sp), 2 * total + 8 * nrec),
        "count4": (lambda: lib.skh_count_dev(ctx, p(seq), p(offs), nrec, total, cls, 4, p(counts), sp), total + 40 * nrec),
        "stats": (lambda: lib.skh_stats_dev(ctx, p(seq), p(offs), nrec, total, gap, p(stats), sp), total + 8 * nrec),
    }
    print("== %s: %d records, %.2f Gbases, max %d" % (name, nrec, total / 1e9, lengths.max()))
    only = os.environ.get("OPS")
    for op, (fn, nbytes) in ops.items():
        if only and op not in only.split(","): continue
        for _ in range(2): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(5): fn()
        e1.record(st)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print("   %-8s %9.3f ms %8.1f GB/s %5.1f%% of 8 TB/s %8.1f Gbases/s" % (op, ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / 80, total / ms / 1e6))
    if only:            # a subset of the operations ran: nothing to cross-check
        del seq, out, back, work
        torch.cuda.empty_cache()
        return True
    # properties
    lib.skh_revcomp_dev(ctx, p(out), p(back), p(offs), nrec, total, lut, 1, sp)
    torch.cuda.synchronize()
    ok_rt = bool(torch.equal(back[:total], seq[:total]))
    # the in-place form ran 2 + 5 times on `work`: once more gives the batch back, and then equals `out` after one more
    lib.skh_revcomp_inplace_dev(ctx, p(work), None, p(offs), nrec, total, lut, 1, sp)
    torch.cuda.synchronize()
    ok_rt &= bool(torch.equal(work[:total], seq[:total]))
    lib.skh_revcomp_inplace_dev(ctx, p(work), None, p(offs), nrec, total, lut, 1, sp)
    torch.cuda.synchronize()
    ok_rt &= bool(torch.equal(work[:total], out[:total]))
    c = counts.view(nrec, 4)
    acgt = sum(int(((seq[:total] == x) | (seq[:total] == x + 32)).sum()) for x in b"ACGT")
    ok_cnt = int(c.sum()) == acgt
    s = [int(x) for x in stats.cpu()]
    gaps = int(((seq[:total] == ord("-")) | (seq[:total] == ord("."))).sum())
    ok_st = s == [nrec, total, gaps, int(lengths.min()), int(lengths.max())]
    # first/last record spot check against numpy
    r0 = out[:int(offs_h[1])].cpu().numpy(); i0 = seq[:int(offs_h[1])].cpu().numpy()
    lut_np = np.frombuffer(bytes(lut), np.uint8)
    ok_spot = bool(np.array_equal(r0, lut_np[i0[::-1]]))
    print("   roundtrip", ok_rt, "counts", ok_cnt, "stats", ok_st, "first-record", ok_spot)
    del seq, out, back, work
    torch.cuda.empty_cache()
    return ok_rt and ok_cnt and ok_st and ok_spot


ok = True
which = os.environ.get("WHICH", "chrom,ont")
if "chrom" in which:
    ok &= run("config5: 24 chromosome-length records", GRCH38, 0xC5)
if "ont" in which:
    rng = np.random.default_rng(0xC4)
    lens = np.clip(np.exp(rng.normal(np.log(8000), 1.1, 700_000)), 1000, 1_000_000).astype(np.int64)
    ok &= run("config4: ONT-like log-normal lengths", lens, 0xC4)
if "trim" in which:      # adapter-trimmed Illumina reads
    rng = np.random.default_rng(7)
    ok &= run("trimmed reads: uniform 30..151", rng.integers(30, 152, 8_000_000), 7)
if "r150" in which:
    ok &= run("fixed 150 bp", np.full(int(os.environ.get("R150_N", 10_000_000)), 150), 12)
if "r100" in which:
    ok &= run("fixed 100 bp", np.full(10_000_000, 100), 8)
if "r250" in which:
    ok &= run("fixed 250 bp", np.full(5_000_000, 250), 9)
for tag, n, L in (("r350", 4_000_000, 350), ("r500", 4_000_000, 500), ("r700", 3_000_000, 700), ("r1k", 2_000_000, 1000), ("r4k", 500_000, 4000), ("r16k", 125_000, 16000)):
    if tag in which:
        rng = np.random.default_rng(L)
        ok &= run("lengths uniform in %d +- 20 %%" % L, rng.integers(int(L * 0.8), int(L * 1.2), n), L)
if "mirna" in which:     # hairpin.fa-like: 39..2354, mostly ~100
    rng = np.random.default_rng(10)
    ok &= run("hairpin-like log-normal around 100", np.clip(np.exp(rng.normal(np.log(100), 0.35, 8_000_000)), 39, 2354).astype(np.int64), 10)
if "mixed" in which:     # short reads with a few long contigs in between
    rng = np.random.default_rng(11)
    lens = np.full(4_000_000, 150); lens[rng.integers(0, len(lens), 200)] = rng.integers(100_000, 3_000_000, 200)
    ok &= run("150 bp reads + 200 long contigs", lens, 11)
if "kbtail" in which:    # heavy-tailed kilobase lengths: steps of the flat kernel overflow its stage now and then
    rng = np.random.default_rng(13)
    ok &= run("heavy-tailed log-normal around 1500 (200..30000)", np.clip(np.exp(rng.normal(np.log(1500), 0.8, 1_000_000)), 200, 30000).astype(np.int64), 13)
if "mix12k" in which:    # nine 150-base reads to one 12-kb record
    rng = np.random.default_rng(14)
    lens = np.full(1_500_000, 150); lens[rng.random(len(lens)) < 0.1] = 12000
    ok &= run("150 bp reads with 10 % records of 12 kb", lens, 14)
lib.skh_destroy(ctx)
sys.exit(0 if ok else 1)
