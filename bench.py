#!/usr/bin/env python3
"""Benchmark of the SeqKit hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

A step is one pass of the hot path over one resident batch: `seq -r -p` on a FASTQ
batch (bases reversed + complemented, qualities reversed) followed by the `fx2tab -g -B`
per-record base counts (classes G, C, A, T) of the same batch.
Workload = BASELINE.json configs[1]: synthetic 10 M x 150 bp Illumina-like FASTQ
(seed 0xC2 + rank), resident in HBM before the timed region.  With N > 1 every rank
holds its own batch of that size (records are independent: weak scaling, no
collective in the data path).

One JSON line on rank 0: value = bases processed by all ranks per second of the
slowest rank, plus
  roofline / kernels   per-kernel figures from HIP events on the launch stream;
  configs              the other BASELINE configurations measured in the same run: at N = 1 their
                       single-GPU kernels (100 M x 150 bp counts, 24 chromosome-length records,
                       nanopore-like lengths), at N > 1 the same total work cut over the ranks by
                       byte volume (strong scaling; chromosomes cut inside the records);
  stats                the path's one collective: per-GPU `stats` totals all-reduced by the
                       library over RCCL (skh_comm_init_rank + skh_stats_allreduce_dev), timed;
  cpu_baseline         the CPU oracle on a bounded sample, one thread, and the same over all host
                       cores (N = 1 only);
  end_to_end           the `seqkit-hip` driver on the same batch written as a FASTQ file
                       (N = 1 only; never `value`).
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak

GRCH38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717,
          133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285,
          58617616, 64444167, 46709983, 50818468, 156040895, 57227415]


def kernels_sha16():
    with open(os.path.join(ROOT, "seqkit_amd", "csrc", "kernels.hip"), "rb") as fh:
        return hashlib.sha256(fh.read()).hexdigest()[:16]


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC passes (profiles/*_hbm_traffic.json),
    valid for the default workload and only while kernels.hip is the file the passes were run on (the
    json carries its hash); None otherwise."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    for name in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if not name.endswith("_hbm_traffic.json"):
            continue
        try:
            with open(os.path.join(pdir, name)) as fh:
                doc = json.load(fh)
            if doc.get("kernels_sha16") == kernels_sha16():
                by_name = {k.replace(" ", ""): v for k, v in doc["kernels"].items()}     # demanglers differ in spacing
                best = by_name[kernel.replace(" ", "")]["hbm_bytes"]
        except (OSError, KeyError, ValueError):
            pass
    return best


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=700, help="timed steps (default: a timed region of about one second)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nrec", type=int, default=10_000_000, help="records per GPU (default: configs[1])")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU baseline budget per leg (0 = skip)")
    ap.add_argument("--sustain", type=float, default=2.0, help="seconds of the same step run before the warm-up and the timed steps (0 = skip)")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the file-to-file run of the driver")
    ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE configurations")
    ap.add_argument("--no-collective", action="store_true", help="skip the RCCL all-reduce of the stats totals (profiler passes)")
    ap.add_argument("--rehearse", action="store_true",
                    help="N ranks on however many GPUs are visible (ranks share a GPU), gloo process group, no RCCL: "
                         "exercises the N > 1 code path on a one-GPU machine; its numbers mean nothing")
    return ap.parse_args()


def synth_plane(total, seed, dev, letters=b"ACGT", weights=None, p_n=2.5e-5, with_qual=False):
    """Bases iid over `letters` (uniform, or `weights`) with P(N) = p_n, qualities iid in '#'..'I',
    generated on the GPU; planes padded to 16 bytes."""
    import torch
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    padded = (total + 15) & ~15
    seq = torch.zeros(padded, dtype=torch.uint8, device=dev)
    qual = torch.zeros(padded, dtype=torch.uint8, device=dev) if with_qual else None
    lt = torch.tensor(list(letters), dtype=torch.uint8, device=dev)
    w = torch.tensor(weights, device=dev) if weights is not None else None
    step = 1 << 26
    for a in range(0, total, step):
        b = min(total, a + step)
        if w is None:
            idx = torch.randint(0, len(letters), (b - a,), generator=g, device=dev, dtype=torch.uint8).to(torch.int64)
        else:
            idx = torch.multinomial(w, b - a, replacement=True, generator=g)
        chunk = lt[idx]
        if p_n:
            chunk[torch.rand(b - a, generator=g, device=dev) < p_n] = ord("N")
        seq[a:b] = chunk
        if with_qual:
            qual[a:b] = torch.randint(ord("#"), ord("I") + 1, (b - a,), generator=g, device=dev, dtype=torch.uint8)
        del idx, chunk
    return seq, qual


def synth_fastq(nrec, L, seed, dev):
    import torch
    seq, qual = synth_plane(nrec * L, seed, dev, with_qual=True)
    offs = torch.arange(nrec + 1, dtype=torch.int64, device=dev) * L
    return seq, qual, offs


def cpu_baseline(seconds, L, lut, cls):
    """The CPU oracle (plain C restatement) on a bounded sample of the same workload: reverse-complement
    with quality reversal + the 4-class count.  One thread, then the same work over all host cores
    (threads over record ranges: ctypes releases the GIL around the calls)."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import Oracle, make_batch
    so = os.path.join(ROOT, "oracle", "libseqkit_oracle.so")
    if not os.path.exists(so):
        return None, None
    ora = Oracle(C.CDLL(so))
    nrec = 400_000
    seq, qual, offs = make_batch(np.full(nrec, L), seed=0xC2, letters=b"ACGT", with_qual=True)
    lut_np = np.frombuffer(bytes(lut), np.uint8)
    cls_np = np.frombuffer(bytes(cls), np.uint8)

    def one_pass(s, q, o):
        ora.transform(s, q, o, lut_np, True)
        ora.count(s, o, cls_np, 4)

    done, t0 = 0, time.perf_counter()
    while True:
        one_pass(seq, qual, offs)
        done += 1
        el = time.perf_counter() - t0
        if el >= seconds or done >= 1000:
            break
    single = {"value": round(done * nrec * L / el / 1e9, 4), "unit": "Gbases/s", "cores": 1, "kind": "port",
              "sample": "%d passes over %d x %d bp FASTQ records (%.1f s): oracle reverse-complement + quality "
                        "reversal + 4-class count, plain C, 1 thread" % (done, nrec, L, el)}
    # all cores of this process's share of the host: one copy of the sample per thread
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    env = os.environ.get("OMP_NUM_THREADS")
    if env and env.isdigit():
        cores = max(1, min(cores, int(env)))
    parts = [(seq.copy(), qual.copy(), offs) for _ in range(cores)]
    with ThreadPoolExecutor(cores) as pool:
        done, t0 = 0, time.perf_counter()
        while True:
            list(pool.map(lambda p: one_pass(*p), parts))
            done += 1
            el = time.perf_counter() - t0
            if el >= seconds or done >= 1000:
                break
    multi = {"value": round(done * cores * nrec * L / el / 1e9, 4), "unit": "Gbases/s", "cores": cores, "kind": "port",
             "sample": "%d rounds of %d threads, each one pass over its own %d x %d bp records (%.1f s): the same oracle "
                       "work on every core this process may use" % (done, cores, nrec, L, el)}
    return single, multi


def write_fastq_text(path, nrec, L, seq, qual):
    """The resident batch as FASTQ text (four lines per record, names of fixed width), assembled on
    the GPU a million records at a time and written to `path`."""
    import torch
    dev = seq.device
    head, tail = b"@SIM:1:FC:1:", b" 1:N:0:ATCACG\n"
    nd = 9
    w1 = len(head) + nd + len(tail)
    width = w1 + 2 * L + 4
    pow10 = torch.tensor([10 ** (nd - 1 - k) for k in range(nd)], dtype=torch.int64, device=dev)
    with open(path, "wb") as fh:
        for a in range(0, nrec, 1_000_000):
            b = min(nrec, a + 1_000_000)
            t = torch.empty((b - a, width), dtype=torch.uint8, device=dev)
            t[:, :len(head)] = torch.tensor(list(head), dtype=torch.uint8, device=dev)
            ids = torch.arange(a, b, dtype=torch.int64, device=dev)
            t[:, len(head):len(head) + nd] = ((ids[:, None] // pow10) % 10 + ord("0")).to(torch.uint8)
            t[:, len(head) + nd:w1] = torch.tensor(list(tail), dtype=torch.uint8, device=dev)
            t[:, w1:w1 + L] = seq[a * L:b * L].view(b - a, L)
            t[:, w1 + L:w1 + L + 3] = torch.tensor(list(b"\n+\n"), dtype=torch.uint8, device=dev)
            t[:, w1 + L + 3:w1 + 2 * L + 3] = qual[a * L:b * L].view(b - a, L)
            t[:, width - 1] = ord("\n")
            t.cpu().numpy().tofile(fh)
            del t
    return os.path.getsize(path)


def end_to_end(nrec, L, seq, qual):
    """File text in, file text out: `seqkit-hip seq -r -p` (the C++ driver over the C ABI: read, upload,
    GPU parse, kernels, GPU formatting, download, write) on the batch written as a FASTQ file, page
    cache warm, output to /dev/null, process start to exit; the CPU restatement's command-line form
    on the first tenth of the same file beside it.  None when a binary is missing."""
    import shutil
    import tempfile
    drv = os.path.join(ROOT, "seqkit_amd", "seqkit-hip")
    ora = os.path.join(ROOT, "oracle", "seqkit_oracle")
    if not os.path.exists(drv):
        return None
    d = tempfile.mkdtemp(prefix="skh_e2e_")
    try:
        fq, small = os.path.join(d, "batch.fq"), os.path.join(d, "tenth.fq")
        nbytes = write_fastq_text(fq, nrec, L, seq, qual)
        n_small = max(nrec // 10, 1)
        with open(fq, "rb") as a, open(small, "wb") as b:
            b.write(a.read(nbytes // nrec * n_small))
        with open(os.devnull, "wb") as null:
            def wall(argv):
                t0 = time.perf_counter()
                rc = subprocess.call(argv, stdout=null)
                return time.perf_counter() - t0 if rc == 0 else None
            wall([drv, "stats", small])                         # first use of the GPU by another process
            runs = [wall([drv, "seq", "-r", "-p", fq]) for _ in range(3)]
            cpu = wall([ora, "seq", "-r", "-p", small]) if os.path.exists(ora) else None
        if any(r is None for r in runs):
            return None
        best = min(runs)
        out = {"command": "seqkit-hip seq -r -p batch.fq > /dev/null", "value": round(nrec * L / best / 1e9, 3),
               "unit": "Gbases/s", "wall_s": round(best, 4), "input_bytes": nbytes,
               "note": "best of 3, page cache warm, process start to exit, host<->device copies included"}
        if cpu:
            out["cpu_baseline"] = {"command": "oracle seq -r -p (plain C restatement, 1 thread) on the first %d records" % n_small,
                                   "value": round(n_small * L / cpu / 1e9, 3), "unit": "Gbases/s", "wall_s": round(cpu, 4),
                                   "cores": 1, "kind": "port"}
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


class Bench:
    """The library, one context and one launch stream with HIP-event timing on that stream."""

    def __init__(self, lib, ctx, dev):
        import torch
        self.torch, self.lib, self.ctx, self.dev = torch, lib, ctx, dev
        self.stream = torch.cuda.Stream(device=dev)      # non-null handle: the library launches on exactly this stream
        assert self.stream.cuda_stream != 0
        self.sp = C.c_void_p(self.stream.cuda_stream)
        self.lut = lib.pair_table(1)                     # DNAredundant (reader's guess for ACGTN reads)
        self.cls = lib.class_table(["G", "C", "A", "T"])  # fx2tab -g -B AGCT needs G, C and A+T
        self.gap = (C.c_uint8 * 256)()
        for ch in b"- .":
            self.gap[ch] = 1

    @staticmethod
    def p(t):
        return C.c_void_p(t.data_ptr())

    def revcomp(self, seq, out, offs, nrec, total, lut=True):
        self.lib.check(self.lib.skh_revcomp_dev(self.ctx, self.p(seq), self.p(out), self.p(offs), nrec, total,
                                                self.lut if lut else None, 1, self.sp))

    def count(self, seq, offs, nrec, total, counts):
        self.lib.check(self.lib.skh_count_dev(self.ctx, self.p(seq), self.p(offs), nrec, total, self.cls, 4,
                                              self.p(counts), self.sp))

    def stats(self, seq, offs, nrec, total, out):
        self.lib.check(self.lib.skh_stats_dev(self.ctx, self.p(seq), self.p(offs), nrec, total, self.gap, self.p(out), self.sp))

    def time_ms(self, fn, reps=5, warm=2):
        """Average duration of fn() over `reps` launches, HIP events on the launch stream."""
        torch = self.torch
        torch.cuda.synchronize()         # the operands were made on torch's own stream
        for _ in range(warm):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(self.stream)
        for _ in range(reps):
            fn()
        e1.record(self.stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps


def figure(ms, nbytes, bases):
    gbs = nbytes / (ms * 1e-3) / 1e9
    return {"ms": round(ms, 4), "gbases_per_s": round(bases / ms / 1e6, 1), "algorithmic_GBps": round(gbs, 1),
            "frac": round(gbs / HBM_PEAK_GBS, 4)}


def offsets_of(lengths, dev):
    import numpy as np
    import torch
    offs = np.zeros(len(lengths) + 1, np.int64)
    np.cumsum(np.asarray(lengths, np.int64), out=offs[1:])
    return torch.from_numpy(offs).to(dev), int(offs[-1])


def other_configs(b, rank, world, host_xfer=False):
    """BASELINE configs[2], [4] and [3] (kernels only).  One rank: the whole configuration on this GPU.
    Several ranks: the same total work cut by byte volume (seqkit_amd.shard), every rank timing its share;
    rank 0 reports the slowest share (strong scaling)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from seqkit_amd import shard
    dev, out = b.dev, {}
    soft = (b"ACGTacgtNn-.", [.24, .24, .24, .24, .0075, .0075, .0075, .0075, .004, .004, .001, .001])

    def slowest(ms):
        if world == 1:
            return ms
        t = torch.tensor([ms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # configs[2]: 100 M x 150 bp, fx2tab -n -l -g -B AGCT counts; ranks take equal record ranges
    nrec_all, L = 100_000_000, 150
    lo, hi = rank * nrec_all // world, (rank + 1) * nrec_all // world     # = split_by_bytes for equal lengths
    nrec, total = hi - lo, (hi - lo) * L
    seq, _ = synth_plane(total, 0xC3 + rank, dev)
    offs = torch.arange(nrec + 1, dtype=torch.int64, device=dev) * L
    counts = torch.empty(nrec * 4, dtype=torch.int64, device=dev)
    ms = slowest(b.time_ms(lambda: b.count(seq, offs, nrec, total, counts)))
    ok = int(counts.view(nrec, 4).sum()) == int((seq[:total] != ord("N")).sum())
    out["configs[2] count4 100M x 150bp"] = dict(figure(ms, (total + 8 * (nrec + 1) + 32 * nrec) * world, total * world),
                                                 records_per_rank=nrec, verified=ok,
                                                 scaling="strong" if world > 1 else "single GPU")
    del seq, offs, counts
    torch.cuda.empty_cache()

    # configs[4]: ONE batch of 24 records with the GRCh38 chromosome lengths, the same on every rank (same seed);
    # the library's rule (skh_split_plane) cuts it into `world` shares of equal byte volume, inside the records;
    # every rank reverse-complements and counts its share, the pieces are gathered on rank 0, put at their
    # mirrored places [L-b, L-a) and compared with the whole batch done on one GPU
    offs_all, all_total = offsets_of(GRCH38, dev)
    offs_np = offs_all.cpu().numpy().astype(np.uint64)
    cuts = np.zeros(world + 1, np.uint64)
    b.lib.check(b.lib.skh_split_plane(offs_np.ctypes.data_as(C.POINTER(C.c_uint64)), 24, world,
                                      cuts.ctypes.data_as(C.POINTER(C.c_uint64))))
    assert [int(c) for c in cuts] == shard.split_plane(offs_np, world)
    whole, _ = synth_plane(all_total, 0xC5, dev, soft[0], soft[1], p_n=0)
    b0, b1 = int(cuts[rank]), int(cuts[rank + 1])
    pieces = shard.pieces_of_share(offs_np, b0, b1)
    lens = [p[2] - p[1] for p in pieces]
    offs, total = offsets_of(lens, dev)
    nrec = len(lens)
    assert total == b1 - b0
    seq = torch.zeros((total + 15) & ~15, dtype=torch.uint8, device=dev)
    seq[:total] = whole[b0:b1]
    outp = torch.empty_like(seq)
    counts = torch.empty(nrec * 4, dtype=torch.int64, device=dev)
    ms_rev = slowest(b.time_ms(lambda: b.revcomp(seq, outp, offs, nrec, total)))
    ms_cnt = slowest(b.time_ms(lambda: b.count(seq, offs, nrec, total, counts)))
    torch.cuda.synchronize()
    # rank 0: the whole batch on one GPU, and the gathered pieces beside it
    ok, md5 = True, None
    if rank == 0:
        whole_out = torch.empty_like(whole)
        whole_cnt = torch.empty(24 * 4, dtype=torch.int64, device=dev)
        b.revcomp(whole, whole_out, offs_all, 24, all_total)
        b.count(whole, offs_all, 24, all_total, whole_cnt)
        torch.cuda.synchronize()
        joined = torch.zeros_like(whole)
        joined_cnt = torch.zeros(24, 4, dtype=torch.int64, device=dev)
    for src in range(world):
        if rank != 0 and rank != src:
            continue
        sb0, sb1 = int(cuts[src]), int(cuts[src + 1])
        spieces = pieces if src == rank else shard.pieces_of_share(offs_np, sb0, sb1)
        if src == 0:
            part_out, part_cnt = outp[:total], counts.view(nrec, 4)
        elif rank == src:
            for t in (outp[:total].contiguous(), counts.view(nrec, 4).contiguous()):
                dist.send(t.cpu() if host_xfer else t, dst=0)
            continue
        else:
            xdev = "cpu" if host_xfer else dev        # (a gloo rehearsal moves host tensors)
            part_out = torch.empty(sb1 - sb0, dtype=torch.uint8, device=xdev)
            part_cnt = torch.empty((len(spieces), 4), dtype=torch.int64, device=xdev)
            dist.recv(part_out, src=src)
            dist.recv(part_cnt, src=src)
            part_out, part_cnt = part_out.to(dev), part_cnt.to(dev)
        pos = 0
        for i, (r, a, e) in enumerate(spieces):          # piece [a, e) of record r, reversed, belongs at [L-e, L-a)
            s0, L = int(offs_np[r]), int(offs_np[r + 1] - offs_np[r])
            joined[s0 + L - e:s0 + L - a] = part_out[pos:pos + (e - a)]
            joined_cnt[r] += part_cnt[i]
            pos += e - a
    if rank == 0:
        ok = bool(torch.equal(joined[:all_total], whole_out[:all_total])) and bool(torch.equal(joined_cnt.view(-1), whole_cnt))
        if os.environ.get("SKH_BENCH_MD5"):
            md5 = hashlib.md5(joined[:all_total].cpu().numpy().tobytes()).hexdigest()
        del whole_out, joined
    out["configs[4] revcomp 24 chromosomes"] = dict(figure(ms_rev, 2 * all_total + 8 * 24, all_total), verified=ok,
                                                    pieces_per_rank=nrec, bytes_per_rank=total,
                                                    check="one shared batch cut by skh_split_plane; the ranks' pieces gathered on rank 0 "
                                                          "equal the whole batch reverse-complemented on one GPU",
                                                    scaling="strong (cuts inside records)" if world > 1 else "single GPU")
    if md5:
        out["configs[4] revcomp 24 chromosomes"]["output_md5"] = md5
    out["configs[4] count4 24 chromosomes"] = dict(figure(ms_cnt, all_total + 40 * 24, all_total), bytes_per_rank=total)
    del seq, outp, counts, offs, whole
    torch.cuda.empty_cache()

    # configs[3]: nanopore-like lengths (log-normal, 1 kbp - 1 Mbp), stats / revcomp / counts; contiguous
    # record ranges of equal byte volume
    rng = np.random.default_rng(0xC4)
    lens_all = np.clip(np.exp(rng.normal(np.log(8000), 1.1, 700_000)), 1000, 1_000_000).astype(np.int64)
    offs_all = np.zeros(len(lens_all) + 1, np.uint64)
    np.cumsum(lens_all.astype(np.uint64), out=offs_all[1:])
    lo, hi = shard.split_by_bytes(offs_all, world)[rank]
    lens = lens_all[lo:hi]
    offs, total = offsets_of(lens, dev)
    nrec = len(lens)
    all_total, all_nrec = int(offs_all[-1]), len(lens_all)
    seq, _ = synth_plane(total, 0xC4 + rank, dev, soft[0], soft[1], p_n=0)
    outp = torch.empty_like(seq)
    counts = torch.empty(nrec * 4, dtype=torch.int64, device=dev)
    st = torch.zeros(5, dtype=torch.int64, device=dev)
    ms_st = slowest(b.time_ms(lambda: b.stats(seq, offs, nrec, total, st)))
    ms_rev = slowest(b.time_ms(lambda: b.revcomp(seq, outp, offs, nrec, total)))
    ms_cnt = slowest(b.time_ms(lambda: b.count(seq, offs, nrec, total, counts)))
    torch.cuda.synchronize()
    got = [int(x) for x in st.cpu()]
    ok = got[:2] == [nrec, total] and got[3:] == [int(lens.min()), int(lens.max())]
    tag = "configs[3] %%s nanopore-like %.1f Gbases" % (all_total / 1e9)
    out[tag % "stats"] = dict(figure(ms_st, all_total + 8 * all_nrec, all_total), verified=ok, records_per_rank=nrec,
                              bytes_per_rank=total, scaling="strong" if world > 1 else "single GPU")
    out[tag % "revcomp"] = figure(ms_rev, 2 * all_total + 8 * all_nrec, all_total)
    out[tag % "count4"] = figure(ms_cnt, all_total + 40 * all_nrec, all_total)
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        # convenience: re-launch under torch.distributed.run as a child process
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import torch
    import torch.distributed as dist
    import seqkit_amd

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    if args.rehearse:
        local %= torch.cuda.device_count()
        args.no_collective = True
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if args.rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    lib = seqkit_amd.load()          # fails loudly when the HIP library is missing
    ctx = lib.init(local)
    b = Bench(lib, ctx, dev)
    p, sp, stream = b.p, b.sp, b.stream
    nrec, L = args.nrec, args.read_len
    total = nrec * L
    seq, qual, offs = synth_fastq(nrec, L, 0xC2 + rank, dev)
    seq_out, qual_out = torch.empty_like(seq), torch.empty_like(qual)
    counts = torch.empty(nrec * 4, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()

    # seq -r -p on FASTQ = bases reversed + complemented, qualities reversed.  The library runs
    # the two planes as two launches of the same kernel; they are issued separately here so
    # that each can be timed on its own (skh_revcomp_fastq_dev does exactly these two calls).
    def revcomp_bases():
        b.revcomp(seq, seq_out, offs, nrec, total)

    def reverse_quals():
        b.revcomp(qual, qual_out, offs, nrec, total, lut=False)

    def count():
        b.count(seq, offs, nrec, total, counts)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the same step for a longer stretch, BEFORE the timed steps (which may take milliseconds): what the GPU
    # sustains -- and the timed steps then start on a GPU at the clocks of a busy one, the state the figure is for,
    # not on one that has been idle while the batch was made
    sustained = None
    if args.sustain > 0:
        done, t_s = 0, time.perf_counter()
        while True:
            for _ in range(50):
                revcomp_bases(); reverse_quals(); count()
            done += 50
            torch.cuda.synchronize()
            el = time.perf_counter() - t_s
            if el >= args.sustain:
                break
        if world > 1:
            t = torch.tensor([done / el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            rate = float(t.item())
        else:
            rate = done / el
        sustained = {"seconds": round(el, 3), "steps": done, "gbases_per_s": round(world * total * rate / 1e9, 3),
                     "ms_per_step": round(1e3 / rate, 4), "when": "before the warm-up and the timed steps"}
    for _ in range(args.warmup):
        revcomp_bases(); reverse_quals(); count()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record(stream)
        revcomp_bases()
        ev[k][1].record(stream)
        reverse_quals()
        ev[k][2].record(stream)
        count()
        ev[k][3].record(stream)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    rev_ms = sum(e[0].elapsed_time(e[1]) for e in ev) / args.steps
    qual_ms = sum(e[1].elapsed_time(e[2]) for e in ev) / args.steps
    cnt_ms = sum(e[2].elapsed_time(e[3]) for e in ev) / args.steps

    # `stats` totals: per-GPU partial on the device, then the path's one collective, done by the
    # library over RCCL (one rank per process; the id travels over torch.distributed's own channel)
    stats_d = torch.zeros(5, dtype=torch.int64, device=dev)
    stats_ms = b.time_ms(lambda: b.stats(seq, offs, nrec, total, stats_d), reps=10)
    allreduce_us, comm_init_s, collective_error, rccl = [], None, None, None
    stats = [int(x) for x in stats_d.cpu()]
    if not args.no_collective:
        # (a failure here is reported in the line, not allowed to take the line with it: the timed steps are done)
        try:
            idbuf = torch.zeros(128, dtype=torch.uint8, device=dev)
            if rank == 0:
                raw = (C.c_uint8 * 128)()
                lib.check(lib.skh_comm_unique_id(raw))
                idbuf.copy_(torch.tensor(list(raw), dtype=torch.uint8))
            if world > 1:
                dist.broadcast(idbuf, src=0)
            raw = (C.c_uint8 * 128)(*[int(x) for x in idbuf.cpu()])
            comm = C.c_void_p()
            t_init = time.perf_counter()
            lib.check(lib.skh_comm_init_rank(ctx, raw, world, rank, C.byref(comm)))
            comm_init_s = time.perf_counter() - t_init
            dparts, streams = (C.c_void_p * 1)(stats_d.data_ptr()), (C.c_void_p * 1)(stream.cuda_stream)
            for it in range(4):                                   # the first call sets up RCCL's channels
                b.stats(seq, offs, nrec, total, stats_d)
                barrier()
                ta = time.perf_counter()
                lib.check(lib.skh_stats_allreduce_dev(comm, dparts, streams))
                torch.cuda.synchronize()
                allreduce_us.append((time.perf_counter() - ta) * 1e6)
            stats = [int(x) for x in stats_d.cpu()]
            if not (stats[:2] == [nrec * world, total * world] and stats[3:] == [L, L]):
                raise RuntimeError("all-reduced totals are wrong: %r" % (stats,))
            lib.skh_comm_destroy(comm)
            v = [C.c_int(0), C.c_int(0), C.c_int(0)]
            lib.check(lib.skh_comm_rccl_version(C.byref(v[0]), C.byref(v[1]), C.byref(v[2])))
            rccl = {"ncclGetVersion": v[0].value, "header_NCCL_VERSION_CODE": v[1].value,
                    "library": "the copy the process held (torch's)" if v[2].value else "opened by libseqkit_hip.so"}
        except Exception as exc:                                  # noqa: BLE001 -- whatever it is, it goes into the line
            collective_error = "%s: %s" % (type(exc).__name__, exc)
            print("[bench] rank %d: collective leg failed: %s" % (rank, collective_error), file=sys.stderr, flush=True)

    # the same reverse-complement with one plane (skh_revcomp_inplace_dev), outside the timed steps
    work = seq.clone()
    inplace_ms = b.time_ms(lambda: lib.check(lib.skh_revcomp_inplace_dev(ctx, p(work), None, p(offs), nrec, total, b.lut, 1, sp)),
                           reps=10)
    del work

    ok = True
    if not args.no_verify:
        # size-independent checks at full size: revcomp is an involution; counts add up
        back = torch.empty_like(seq)
        b.revcomp(seq_out, back, offs, nrec, total)
        torch.cuda.synchronize()
        ok = bool(torch.equal(back[:total], seq[:total])) and \
            int(counts.sum()) == int((seq[:total] != ord("N")).sum()) and \
            bool(torch.equal(qual_out[:total].view(nrec, L).flip(1), qual[:total].view(nrec, L)))
        del back
    if world > 1:
        f = torch.tensor([int(ok)], device=dev)
        dist.all_reduce(f, op=dist.ReduceOp.MIN)
        ok = bool(f.item())

    configs = None
    if not args.no_configs:
        keep = (seq, qual)
        if world > 1:                                    # every GPU needs the room; the e2e leg (N = 1) needs the batch
            del seq_out, qual_out, counts
            torch.cuda.empty_cache()
        try:
            configs = other_configs(b, rank, world, host_xfer=args.rehearse)
        except Exception as exc:                                  # noqa: BLE001 -- reported, the headline figure stands
            configs = {"error": "%s: %s" % (type(exc).__name__, exc)}
            print("[bench] rank %d: configs leg failed: %s" % (rank, configs["error"]), file=sys.stderr, flush=True)
        seq, qual = keep

    if rank == 0:
        # the kernel the library picked for this batch shape (its own selection, asked for, not assumed)
        kname = C.create_string_buffer(128)
        lib.check(lib.skh_kernel_for(ctx, seqkit_amd.capi.OP_REVCOMP, nrec, total, 1, kname, 128))
        rev_kernel = kname.value.decode()
        ms = elapsed / args.steps * 1e3
        value = world * total * args.steps / elapsed / 1e9
        rev_bytes = 2.0 * total + 8.0 * (nrec + 1)          # one plane read + written, offsets read
        cnt_bytes = 1.0 * total + 8.0 * (nrec + 1) + 32.0 * nrec
        st_bytes = 1.0 * total + 8.0 * (nrec + 1)
        achieved = rev_bytes / (rev_ms * 1e-3) / 1e9
        out = {
            "metric": "Gbases/sec revcomp+base-count (kernel & end-to-end), 1-8 MI355X; % HBM peak",
            "value": round(value, 3), "unit": "Gbases/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "configs[1]: %d x %d bp synthetic Illumina FASTQ per GPU, seq -r -p "
                                   "(reverse-complement + quality reversal) then fx2tab -g -B AGCT counts, "
                                   "resident in HBM" % (nrec, L),
                       "records_per_gpu": nrec, "read_len": L, "parallelism": "records sharded, %d rank(s)" % world},
            "roofline": {"bound": "hbm", "kernel": rev_kernel + " (bases: reverse + complement, two planes)",
                         "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": measured_traffic(rev_kernel)
                         if (nrec, L) == (10_000_000, 150) else None,
                         "traffic_source": "committed rocprofv3 --pmc passes under profiles/ whose kernels.hip hash (%s) "
                                           "matches this tree; null otherwise" % kernels_sha16(),
                         "algorithmic_bytes_per_launch": rev_bytes,
                         "avg_launch_ms": round(rev_ms, 4)},
            "kernels": {
                "revcomp_bases": figure(rev_ms, rev_bytes, total),
                "reverse_quals": figure(qual_ms, rev_bytes, total),
                "seq_r_p_fastq": figure(rev_ms + qual_ms, 2 * rev_bytes, total),
                "revcomp_bases_in_place": figure(inplace_ms, rev_bytes, total),
                "count4": figure(cnt_ms, cnt_bytes, total),
                "stats": figure(stats_ms, st_bytes, total),
            },
            "stats": {"totals": stats, "collective": "skh_stats_allreduce_dev: RCCL all-reduce of 5 x int64 (sum x3, min, max) "
                                                     "inside libseqkit_hip.so, %d rank(s)" % world,
                      "allreduce_us": [round(x, 1) for x in allreduce_us],
                      "comm_init_s": None if comm_init_s is None else round(comm_init_s, 3), "rccl": rccl,
                      "error": collective_error},
            "verified": ok,
        }
        if sustained is not None:
            out["sustained"] = sustained
        if configs is not None:
            out["configs"] = configs
        if args.cpu_seconds > 0 and world == 1:
            try:
                single, multi = cpu_baseline(args.cpu_seconds, L, b.lut, b.cls)
                out["cpu_baseline"] = single
                out["cpu_baseline_all_cores"] = multi
            except Exception as exc:                              # noqa: BLE001
                out["cpu_baseline"] = {"value": None, "error": "%s: %s" % (type(exc).__name__, exc)}
        if not args.no_e2e and world == 1:
            torch.cuda.synchronize()
            try:
                out["end_to_end"] = end_to_end(nrec, L, seq, qual)
            except (OSError, subprocess.SubprocessError) as exc:      # e.g. no room for the file: the line still counts
                out["end_to_end"] = {"value": None, "error": str(exc)}
        print(json.dumps(out), flush=True)
    lib.skh_destroy(ctx)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        sys.exit(1)


if __name__ == "__main__":
    main()
