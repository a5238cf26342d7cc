"""GPU parity at the full sizes of the benchmark configurations: planes beyond 4 GiB (a
100 M x 150 bp FASTQ batch is a 15 GB plane), chromosome-length records, nanopore-like length
skew.  The CPU oracle cannot cover 15-30 GB in seconds, so every output byte is checked against
an independent formulation written with plain torch tensor operations (flip / index / compare
on record views), the oracle itself is run on samples of records taken around every 4-GiB
multiple of the plane, and the involution property closes the loop."""
import ctypes as C

import numpy as np
import pytest

from helpers import Oracle, ptr

pytestmark = pytest.mark.gpu

DEV = "cuda:0"

# GRCh38 primary assembly, chr1..22, X, Y (sum 3,088,269,832)
GRCH38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636,
          138394717, 133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345,
          83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415]


@pytest.fixture(scope="module")
def lib():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from seqkit_amd import load
    return load()


@pytest.fixture(scope="module")
def ctx(lib):
    c = lib.init(0)
    yield c
    lib.skh_destroy(c)


@pytest.fixture(scope="module")
def ora(oracle_lib):
    return Oracle(oracle_lib)


def random_plane(total, letters, weights, seed):
    """`total` bytes drawn from `letters` with the given weights, padded to 16 bytes (pad = 0)."""
    import torch
    g = torch.Generator(device=DEV)
    g.manual_seed(seed)
    padded = (total + 15) & ~15
    plane = torch.zeros(padded, dtype=torch.uint8, device=DEV)
    lt = torch.tensor(list(letters), dtype=torch.uint8, device=DEV)
    cdf = torch.tensor(np.cumsum(np.asarray(weights, np.float64) / np.sum(weights)), dtype=torch.float32, device=DEV)
    step = 1 << 28
    for a in range(0, total, step):
        b = min(total, a + step)
        u = torch.rand(b - a, generator=g, device=DEV)
        idx = torch.bucketize(u, cdf).clamp_(max=len(letters) - 1)
        plane[a:b] = lt[idx]
        del u, idx
    return plane


def lut_tensor(lut):
    import torch
    return torch.tensor(list(bytes(lut)), dtype=torch.uint8, device=DEV)


def sample_records_against_oracle(ora, offs_np, seq, out, counts, lut_np, cls_np, nclass, records):
    """Runs the CPU oracle on the given records (copied to the host one by one)."""
    for r in records:
        a, b = int(offs_np[r]), int(offs_np[r + 1])
        n = b - a
        buf = np.zeros(max((n + 15) & ~15, 16), np.uint8)
        buf[:n] = seq[a:b].cpu().numpy()
        o = np.array([0, n], np.uint64)
        if out is not None:
            want, _ = ora.transform(buf, None, o, lut_np, True)
            assert np.array_equal(out[a:b].cpu().numpy(), want[:n]), "record %d" % r
        if counts is not None:
            want = ora.count(buf, o, cls_np, nclass)[0]
            got = counts[r * nclass:(r + 1) * nclass].cpu().numpy().view(np.uint64)
            assert np.array_equal(got, want), "record %d" % r


def records_around_4gib_multiples(offs_np, span=3):
    nrec = len(offs_np) - 1
    pick = set(range(min(span, nrec))) | set(range(max(nrec - span, 0), nrec))
    k = 1
    while (k << 32) < int(offs_np[-1]):
        r = int(np.searchsorted(offs_np, np.uint64(k << 32), side="right")) - 1
        pick |= {x for x in range(r - span, r + span + 1) if 0 <= x < nrec}
        k += 1
    return sorted(pick)


def test_100m_reads_counts_and_revcomp(lib, ctx, ora):
    # 100 M x 150 bp: fx2tab -n -l -g -B AGCT counts (and seq -r -p) over a 15 GB plane
    import torch
    nrec, L = 100_000_000, 150
    total = nrec * L
    seq = random_plane(total, b"ACGTNacgt", [24, 24, 24, 24, 1, 1, 1, 0.5, 0.5], seed=0xC3)
    offs = torch.arange(nrec + 1, dtype=torch.int64, device=DEV) * L
    offs_np = np.arange(nrec + 1, dtype=np.uint64) * np.uint64(L)
    cls = lib.class_table(["G", "C", "A", "T"])
    counts = torch.full((nrec * 4,), -1, dtype=torch.int64, device=DEV)
    out = torch.full_like(seq, 0xEE)
    lut = lib.pair_table(1)
    torch.cuda.synchronize()
    lib.check(lib.skh_count_dev(ctx, ptr(seq), ptr(offs), nrec, total, cls, 4, ptr(counts), None))
    lib.check(lib.skh_revcomp_dev(ctx, ptr(seq), ptr(out), ptr(offs), nrec, total, lut, 1, None))
    lib.check(lib.skh_sync(ctx))
    c = counts.view(nrec, 4)
    lt = lut_tensor(lut)
    step = 2_000_000
    for a in range(0, nrec, step):
        b = min(nrec, a + step)
        v = seq[a * L:b * L].view(b - a, L)
        for k, (up, lo) in enumerate((b"Gg", b"Cc", b"Aa", b"Tt")):
            want = ((v == up) | (v == lo)).sum(1)
            assert torch.equal(c[a:b, k], want), "class %d, records %d.." % (k, a)
        want = lt[v.flip(1).to(torch.int32)]
        assert torch.equal(out[a * L:b * L].view(b - a, L), want), "records %d.." % a
        del v, want
    assert bool((out[total:] == 0xEE).all())
    sample_records_against_oracle(ora, offs_np, seq, out, counts, np.frombuffer(bytes(lut), np.uint8),
                                  np.frombuffer(bytes(cls), np.uint8), 4, records_around_4gib_multiples(offs_np))
    # stats totals of the same batch
    stats = torch.zeros(5, dtype=torch.int64, device=DEV)
    is_gap = (C.c_uint8 * 256)()
    for ch in b"- .N":
        is_gap[ch] = 1
    torch.cuda.synchronize()
    lib.check(lib.skh_stats_dev(ctx, ptr(seq), ptr(offs), nrec, total, is_gap, ptr(stats), None))
    lib.check(lib.skh_sync(ctx))
    n_count = sum(int((seq[a:min(total, a + (1 << 30))] == ord("N")).sum()) for a in range(0, total, 1 << 30))
    assert [int(x) for x in stats.cpu()] == [nrec, total, n_count, L, L]


def chromosome_batch(seed):
    """24 records with the GRCh38 chromosome lengths: upper case, soft-masked lower-case runs,
    N blocks at the ends and in the middle of every record, rare IUPAC codes."""
    import torch
    lens = np.array(GRCH38, np.uint64)
    offs_np = np.zeros(len(lens) + 1, np.uint64)
    np.cumsum(lens, out=offs_np[1:])
    total = int(offs_np[-1])
    seq = random_plane(total, b"ACGTRYKMSWBDHV", [250] * 4 + [1] * 10, seed)
    # soft-masked runs: about half of the plane in lower case, in runs of 2^12 bytes
    g = torch.Generator(device=DEV)
    g.manual_seed(seed + 1)
    run = 1 << 12
    nrun = (total + run - 1) // run
    mask = (torch.rand(nrun, generator=g, device=DEV) < 0.5)
    for a in range(0, nrun, 1 << 16):
        b = min(nrun, a + (1 << 16))
        lo, hi = a * run, min(total, b * run)
        m = mask[a:b].repeat_interleave(run)[:hi - lo]
        seq[lo:hi] |= (m.to(torch.uint8) << 5)
        del m
    for r in range(len(lens)):
        a, b = int(offs_np[r]), int(offs_np[r + 1])
        seq[a:a + 10_000] = ord("N")
        seq[b - 10_007:b] = ord("N")
        mid = (a + b) // 2 + r
        seq[mid:mid + 3_000_001] = ord("N")
    return seq, offs_np, total


def test_chromosome_records_revcomp_with_intra_record_split(lib, ctx, ora):
    # 24 chromosome-length records (3.1 Gbases), whole and cut into 8 ranks' shares
    import torch
    from seqkit_amd import shard
    seq, offs_np, total = chromosome_batch(0xC5)
    nrec = len(offs_np) - 1
    offs = torch.from_numpy(offs_np.view(np.int64)).to(DEV)
    out = torch.full_like(seq, 0xEE)
    lut = lib.pair_table(1)
    cls = lib.class_table(["G", "C", "A", "T", "N"])
    counts = torch.full((nrec * 5,), -1, dtype=torch.int64, device=DEV)
    torch.cuda.synchronize()
    lib.check(lib.skh_revcomp_dev(ctx, ptr(seq), ptr(out), ptr(offs), nrec, total, lut, 1, None))
    lib.check(lib.skh_count_dev(ctx, ptr(seq), ptr(offs), nrec, total, cls, 5, ptr(counts), None))
    lib.check(lib.skh_sync(ctx))
    lt = lut_tensor(lut)
    c = counts.view(nrec, 5)
    for r in range(nrec):
        a, b = int(offs_np[r]), int(offs_np[r + 1])
        want = lt[seq[a:b].flip(0).to(torch.int32)]
        assert torch.equal(out[a:b], want), "record %d" % r
        del want
        for k, (up, lo) in enumerate((b"Gg", b"Cc", b"Aa", b"Tt", b"Nn")):
            assert int(c[r, k]) == int(((seq[a:b] == up) | (seq[a:b] == lo)).sum()), (r, k)
    assert bool((out[total:] == 0xEE).all())
    # the two shortest records through the CPU oracle as well
    lut_np, cls_np = np.frombuffer(bytes(lut), np.uint8), np.frombuffer(bytes(cls), np.uint8)
    sample_records_against_oracle(ora, offs_np, seq, out, counts, lut_np, cls_np, 5, [20, 21])

    # eight ranks: records are cut inside (slice [a, b) of a record lands at [L-b, L-a) of its
    # output) so that no piece exceeds a quarter of a rank's share, then the pieces are dealt out
    # as contiguous ranges of equal byte volume
    nparts = 8
    share = total // nparts // 4
    pieces = []                                   # (in_pos, length, out_pos)
    for r in range(nrec):
        s, L = int(offs_np[r]), int(offs_np[r + 1] - offs_np[r])
        k = max(1, -(-L // share))
        for a, b, opos in shard.split_record(L, k):
            pieces.append((s + a, b - a, s + opos))
    assert len(pieces) > nrec
    vol = np.array([p[1] for p in pieces], np.uint64)
    poffs = np.zeros(len(pieces) + 1, np.uint64)
    np.cumsum(vol, out=poffs[1:])
    out2 = torch.full_like(seq, 0xEE)
    for lo, hi in shard.split_by_bytes(poffs, nparts):
        if lo == hi:
            continue
        # a rank's share as a batch of its own: its slices packed back to back
        n = int(poffs[hi] - poffs[lo])
        r_in = torch.zeros((n + 15) & ~15, dtype=torch.uint8, device=DEV)
        r_out = torch.full_like(r_in, 0xEE)
        for p in range(lo, hi):
            ip, ln, _ = pieces[p]
            q = int(poffs[p] - poffs[lo])
            r_in[q:q + ln] = seq[ip:ip + ln]
        r_offs = torch.from_numpy((poffs[lo:hi + 1] - poffs[lo]).view(np.int64)).to(DEV)
        torch.cuda.synchronize()
        lib.check(lib.skh_revcomp_dev(ctx, ptr(r_in), ptr(r_out), ptr(r_offs), hi - lo, n, lut, 1, None))
        lib.check(lib.skh_sync(ctx))
        for p in range(lo, hi):
            _, ln, op = pieces[p]
            q = int(poffs[p] - poffs[lo])
            out2[op:op + ln] = r_out[q:q + ln]
        del r_in, r_out
    assert torch.equal(out2, out)


def test_nanopore_like_lengths_stats_revcomp_counts(lib, ctx, ora):
    # ONT-like reads: log-normal lengths clipped to 1 kbp .. 1 Mbp, 30 Gbases, gap letters at 0.1 %
    import torch
    rng = np.random.default_rng(0xC4)
    lens = np.clip(rng.lognormal(np.log(8000.0), 1.1, 2_400_000), 1000, 1_000_000).astype(np.uint64)
    offs_np = np.zeros(len(lens) + 1, np.uint64)
    np.cumsum(lens, out=offs_np[1:])
    nrec = int(np.searchsorted(offs_np, np.uint64(30_000_000_000), side="left"))
    assert nrec < len(lens)
    lens, offs_np = lens[:nrec], offs_np[:nrec + 1]
    total = int(offs_np[-1])
    seq = random_plane(total, b"ACGTN-.", [249.5] * 4 + [1, 0.5, 0.5], seed=0xC4)
    offs = torch.from_numpy(offs_np.view(np.int64)).to(DEV)

    stats = torch.zeros(5, dtype=torch.int64, device=DEV)
    is_gap = (C.c_uint8 * 256)()
    for ch in b"- .":
        is_gap[ch] = 1
    torch.cuda.synchronize()
    lib.check(lib.skh_stats_dev(ctx, ptr(seq), ptr(offs), nrec, total, is_gap, ptr(stats), None))
    lib.check(lib.skh_sync(ctx))
    gaps = 0
    for a in range(0, total, 1 << 30):
        v = seq[a:min(total, a + (1 << 30))]
        gaps += int(((v == ord("-")) | (v == ord(".")) | (v == ord(" "))).sum())
    assert [int(x) for x in stats.cpu()] == [nrec, total, gaps, int(lens.min()), int(lens.max())]
    assert gaps > 0

    out = torch.full_like(seq, 0xEE)
    back = torch.empty_like(seq)
    lut = lib.pair_table(1)
    cls = lib.class_table(["G", "C", "A", "T", "-. "])
    counts = torch.full((nrec * 5,), -1, dtype=torch.int64, device=DEV)
    torch.cuda.synchronize()          # torch filled the buffers on its own stream
    lib.check(lib.skh_revcomp_dev(ctx, ptr(seq), ptr(out), ptr(offs), nrec, total, lut, 1, None))
    lib.check(lib.skh_revcomp_dev(ctx, ptr(out), ptr(back), ptr(offs), nrec, total, lut, 1, None))
    lib.check(lib.skh_count_dev(ctx, ptr(seq), ptr(offs), nrec, total, cls, 5, ptr(counts), None))
    lib.check(lib.skh_sync(ctx))
    assert torch.equal(back[:total], seq[:total])
    assert bool((out[total:] == 0xEE).all())
    del back
    # per-record sums of the counts against segment sums of the class indicator planes
    c = counts.view(nrec, 5)
    assert int(c[:, 4].sum()) == gaps
    csum = c.sum(1)
    n_per_rec = torch.from_numpy(lens.astype(np.int64)).to(DEV) - csum          # what is left are the N bytes
    assert int(n_per_rec.min()) >= 0
    n_total = sum(int((seq[a:min(total, a + (1 << 30))] == ord("N")).sum()) for a in range(0, total, 1 << 30))
    assert int(n_per_rec.sum()) == n_total
    # the records around every 4-GiB multiple, the longest and the shortest, through the CPU oracle
    pick = set(records_around_4gib_multiples(offs_np, span=1))
    pick |= {int(np.argmax(lens)), int(np.argmin(lens))}
    lut_np, cls_np = np.frombuffer(bytes(lut), np.uint8), np.frombuffer(bytes(cls), np.uint8)
    sample_records_against_oracle(ora, offs_np, seq, out, counts, lut_np, cls_np, 5, sorted(pick))
    # every record's output against flip + table lookup, for a spread of 2000 records
    lt = lut_tensor(lut)
    for r in np.linspace(0, nrec - 1, 2000).astype(np.int64):
        a, b = int(offs_np[r]), int(offs_np[r + 1])
        assert torch.equal(out[a:b], lt[seq[a:b].flip(0).to(torch.int32)]), "record %d" % r
