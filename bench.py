#!/usr/bin/env python3
"""Benchmark of the SeqKit hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

A step is one pass of the hot path over one resident batch: `seq -r -p` on a FASTQ
batch (bases reversed + complemented, qualities reversed, one fused launch) followed
by the `fx2tab -g -B` per-record base counts (classes G, C, A, T) of the same batch.
Workload = BASELINE.json configs[1]: synthetic 10 M x 150 bp Illumina-like FASTQ
(seed 0xC2 + rank), resident in HBM before the timed region.  With N > 1 every rank
holds its own batch of that size (records are independent: weak scaling, no
collective in the data path); the `stats` totals are all-reduced once over RCCL after
the timed loop and checked.

One JSON line on rank 0: value = bases processed by all ranks per second of the
slowest rank, plus per-kernel roofline figures from HIP events on the launch stream,
the CPU oracle timed on a bounded sample and, at N = 1, the end-to-end rate of the
`seqkit-hip` driver on the same batch written as a FASTQ file (`end_to_end`, never `value`).
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/), valid
    for the default workload only; None otherwise."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")) as fh:
            return json.load(fh)["kernels"][kernel]["hbm_bytes"]
    except (OSError, KeyError, ValueError):
        return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nrec", type=int, default=10_000_000, help="records per GPU (default: configs[1])")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU baseline budget (0 = skip)")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the file-to-file run of the driver")
    return ap.parse_args()


def synth_fastq(nrec, L, seed, dev):
    """Illumina-like planes generated on the GPU: bases iid over ACGT with P(N) = 2.5e-5,
    qualities iid in '#'..'I'."""
    import torch
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    total = nrec * L
    padded = (total + 15) & ~15
    seq = torch.zeros(padded, dtype=torch.uint8, device=dev)
    qual = torch.zeros(padded, dtype=torch.uint8, device=dev)
    letters = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    step = 1 << 26
    for a in range(0, total, step):
        b = min(total, a + step)
        idx = torch.randint(0, 4, (b - a,), generator=g, device=dev, dtype=torch.uint8)
        chunk = letters[idx.to(torch.int64)]
        n_mask = torch.rand(b - a, generator=g, device=dev) < 2.5e-5
        chunk[n_mask] = ord("N")
        seq[a:b] = chunk
        qual[a:b] = torch.randint(ord("#"), ord("I") + 1, (b - a,), generator=g, device=dev, dtype=torch.uint8)
        del idx, chunk, n_mask
    offs = torch.arange(nrec + 1, dtype=torch.int64, device=dev) * L
    return seq, qual, offs


def cpu_baseline(seconds, L, lut, cls):
    """The CPU oracle (plain C restatement, one thread) on a bounded sample of the same
    workload: reverse-complement with quality reversal + the 4-class count."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import Oracle, make_batch
    so = os.path.join(ROOT, "oracle", "libseqkit_oracle.so")
    if not os.path.exists(so):
        return None
    ora = Oracle(C.CDLL(so))
    nrec = 400_000
    seq, qual, offs = make_batch(np.full(nrec, L), seed=0xC2, letters=b"ACGT", with_qual=True)
    lut_np = np.frombuffer(bytes(lut), np.uint8)
    cls_np = np.frombuffer(bytes(cls), np.uint8)
    done, t0 = 0, time.perf_counter()
    while True:
        ora.transform(seq, qual, offs, lut_np, True)
        ora.count(seq, offs, cls_np, 4)
        done += 1
        el = time.perf_counter() - t0
        if el >= seconds or done >= 1000:
            break
    bases = done * nrec * L
    return {"value": round(bases / el / 1e9, 4), "unit": "Gbases/s", "cores": 1, "kind": "port",
            "sample": "%d passes over %d x %d bp FASTQ records (%.1f s): oracle reverse-complement + quality "
                      "reversal + 4-class count, plain C, 1 thread" % (done, nrec, L, el)}


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
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    lib = seqkit_amd.load()          # fails loudly when the HIP library is missing
    ctx = lib.init(local)
    nrec, L = args.nrec, args.read_len
    total = nrec * L
    seq, qual, offs = synth_fastq(nrec, L, 0xC2 + rank, dev)
    seq_out, qual_out = torch.empty_like(seq), torch.empty_like(qual)
    counts = torch.empty(nrec * 4, dtype=torch.int64, device=dev)
    lut = lib.pair_table(1)                          # DNAredundant (reader's guess for ACGTN reads)
    cls = lib.class_table(["G", "C", "A", "T"])      # fx2tab -g -B AGCT needs G, C and A+T
    # a stream of our own with a non-null handle: the library launches on exactly this
    # stream and the HIP events below are recorded on it
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    sp = C.c_void_p(stream.cuda_stream)
    assert stream.cuda_stream != 0
    p = lambda t: C.c_void_p(t.data_ptr())

    # seq -r -p on FASTQ = bases reversed + complemented, qualities reversed.  The library runs
    # the two planes as two launches of the same kernel; they are issued separately here so
    # that each can be timed on its own (skh_revcomp_fastq_dev does exactly these two calls).
    def revcomp_bases():
        lib.check(lib.skh_revcomp_dev(ctx, p(seq), p(seq_out), p(offs), nrec, total, lut, 1, sp))

    def reverse_quals():
        lib.check(lib.skh_revcomp_dev(ctx, p(qual), p(qual_out), p(offs), nrec, total, None, 1, sp))

    def count():
        lib.check(lib.skh_count_dev(ctx, p(seq), p(offs), nrec, total, cls, 4, p(counts), sp))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

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

    # `stats` totals: per-GPU partial on the device, then the path's one collective
    stats_d = torch.zeros(5, dtype=torch.int64, device=dev)
    is_gap = (C.c_uint8 * 256)()
    for ch in b"- .":
        is_gap[ch] = 1
    torch.cuda.synchronize()
    ts = time.perf_counter()
    lib.check(lib.skh_stats_dev(ctx, p(seq), p(offs), nrec, total, is_gap, p(stats_d), sp))
    torch.cuda.synchronize()
    stats_ms = (time.perf_counter() - ts) * 1e3
    allreduce_us = None
    if world > 1:
        from seqkit_amd import shard
        torch.cuda.synchronize(); ta = time.perf_counter()
        tot = shard.allreduce_stats(stats_d)              # the path's only collective (RCCL)
        torch.cuda.synchronize(); allreduce_us = (time.perf_counter() - ta) * 1e6
        stats = [int(x) for x in tot.cpu()]
    else:
        stats = [int(x) for x in stats_d.cpu()]
    assert stats[:2] == [nrec * world, total * world] and stats[3:] == [L, L], stats

    ok = True
    if not args.no_verify:
        # size-independent checks at full size: revcomp is an involution; counts add up
        back = torch.empty_like(seq)
        lib.check(lib.skh_revcomp_dev(ctx, p(seq_out), p(back), p(offs), nrec, total, lut, 1, sp))
        torch.cuda.synchronize()
        ok = bool(torch.equal(back[:total], seq[:total])) and \
            int(counts.sum()) == int((seq[:total] != ord("N")).sum()) and \
            bool(torch.equal(qual_out[:total].view(nrec, L).flip(1), qual[:total].view(nrec, L)))
        del back
    if world > 1:
        f = torch.tensor([int(ok)], device=dev)
        dist.all_reduce(f, op=dist.ReduceOp.MIN)
        ok = bool(f.item())

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * total * args.steps / elapsed / 1e9
        rev_bytes = 2.0 * total + 8.0 * (nrec + 1)          # one plane read + written, offsets read
        cnt_bytes = 1.0 * total + 8.0 * (nrec + 1) + 32.0 * nrec
        achieved = rev_bytes / (rev_ms * 1e-3) / 1e9
        gbs = lambda nbytes, ms: round(nbytes / (ms * 1e-3) / 1e9, 1)
        out = {
            "metric": "Gbases/sec revcomp+base-count (kernel & end-to-end), 1-8 MI355X; % HBM peak",
            "value": round(value, 3), "unit": "Gbases/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "configs[1]: %d x %d bp synthetic Illumina FASTQ per GPU, seq -r -p "
                                   "(reverse-complement + quality reversal) then fx2tab -g -B AGCT counts, "
                                   "resident in HBM" % (nrec, L),
                       "records_per_gpu": nrec, "read_len": L, "parallelism": "records sharded, %d rank(s)" % world},
            "roofline": {"bound": "hbm", "kernel": "skh::k_reverse_staged<true> (bases: reverse + complement)",
                         "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": measured_traffic("skh::k_reverse_staged<true>")
                         if (nrec, L) == (10_000_000, 150) else None,
                         "algorithmic_bytes_per_launch": rev_bytes,
                         "avg_launch_ms": round(rev_ms, 4)},
            "kernels": {
                "revcomp_bases": {"ms": round(rev_ms, 4), "gbases_per_s": round(total / rev_ms / 1e6, 1),
                                  "algorithmic_GBps": gbs(rev_bytes, rev_ms),
                                  "frac": round(gbs(rev_bytes, rev_ms) / HBM_PEAK_GBS, 4)},
                "reverse_quals": {"ms": round(qual_ms, 4), "algorithmic_GBps": gbs(rev_bytes, qual_ms),
                                  "frac": round(gbs(rev_bytes, qual_ms) / HBM_PEAK_GBS, 4)},
                "seq_r_p_fastq": {"ms": round(rev_ms + qual_ms, 4),
                                  "gbases_per_s": round(total / (rev_ms + qual_ms) / 1e6, 1),
                                  "algorithmic_GBps": gbs(2 * rev_bytes, rev_ms + qual_ms)},
                "count4": {"ms": round(cnt_ms, 4), "gbases_per_s": round(total / cnt_ms / 1e6, 1),
                           "algorithmic_GBps": round(cnt_bytes / (cnt_ms * 1e-3) / 1e9, 1),
                           "frac": round(cnt_bytes / (cnt_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                "stats": {"ms_wall": round(stats_ms, 4), "allreduce_us": allreduce_us, "totals": stats},
            },
            "verified": ok,
        }
        if args.cpu_seconds > 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds, L, lut, cls)
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
