#!/usr/bin/env python3
"""Writes a nanopore-like FASTQ: read lengths log-normal (median 8 kbp, sigma 1.1) clipped to
1 kbp .. 1 Mbp, bases ACGT with 0.1 % gap letters ('-', '.') and a few N, qualities '#'..'I'.
usage: gen_ont.py out.fq TOTAL_BASES [seed]"""
import sys
import numpy as np

out, total = sys.argv[1], int(float(sys.argv[2]))
seed = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0xC4
rng = np.random.default_rng(seed)
letters = np.frombuffer(b"ACGT", np.uint8)
done, i = 0, 0
with open(out, "wb") as fh:
    while done < total:
        lens = np.clip(rng.lognormal(np.log(8000.0), 1.1, 2000), 1000, 1_000_000).astype(np.int64)
        n = int(lens.sum())
        seq = letters[rng.integers(0, 4, n, dtype=np.uint8)]
        k = max(n // 1000, 1)
        seq[rng.integers(0, n, k)] = np.frombuffer(b"-.", np.uint8)[rng.integers(0, 2, k)]
        seq[rng.integers(0, n, k // 10 + 1)] = ord("N")
        qual = rng.integers(ord("#"), ord("I") + 1, n, dtype=np.uint8)
        parts, a = [], 0
        for L in lens:
            parts.append(b"@read_%d runid=0f3a ch=%d start_time=2026-01-01T00:00:00Z\n" % (i, i % 512))
            parts.append(seq[a:a + L].tobytes()); parts.append(b"\n+\n")
            parts.append(qual[a:a + L].tobytes()); parts.append(b"\n")
            a += int(L); i += 1
            done += int(L)
            if done >= total:
                break
        fh.write(b"".join(parts))
