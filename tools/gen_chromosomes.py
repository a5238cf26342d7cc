#!/usr/bin/env python3
"""Writes a human-genome-shaped FASTA: 24 records with the GRCh38 chromosome lengths (optionally
scaled down), 60-column lines, upper-case ACGT with soft-masked (lower-case) runs, N blocks at the
ends and in the middle of every record, rare IUPAC codes.
usage: gen_chromosomes.py out.fa [scale] [seed]"""
import sys
import numpy as np

GRCH38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636,
          138394717, 133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345,
          83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415]
NAMES = [str(i) for i in range(1, 23)] + ["X", "Y"]

out = sys.argv[1]
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
seed = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0xC5
rng = np.random.default_rng(seed)
letters = np.frombuffer(b"ACGT", np.uint8)
iupac = np.frombuffer(b"RYKMSWBDHV", np.uint8)
W = 60
with open(out, "wb") as fh:
    for name, full in zip(NAMES, GRCH38):
        L = max(int(full * scale), 100)
        fh.write(b">chr%s  AC:CM000%s.2  gi:568336%s  LN:%d  rl:Chromosome  M5:0  AS:GRCh38\n"
                 % (name.encode(), name.encode(), name.encode(), L))
        seq = letters[rng.integers(0, 4, L, dtype=np.uint8)]
        k = max(L // 100000, 1)
        seq[rng.integers(0, L, k)] = iupac[rng.integers(0, len(iupac), k)]
        run = 1 << 12                                   # soft-masked runs
        mask = np.repeat(rng.random((L + run - 1) // run) < 0.5, run)[:L]
        seq[mask] |= 0x20
        nb = min(10_000, L // 10)
        seq[:nb] = ord("N"); seq[L - nb - 7:] = ord("N")
        mid = L // 2
        seq[mid:mid + min(3_000_001, L // 50)] = ord("N")
        nl = L // W
        body = np.full((nl, W + 1), ord("\n"), np.uint8)
        body[:, :W] = seq[:nl * W].reshape(nl, W)
        fh.write(body.tobytes())
        if L % W:
            fh.write(seq[nl * W:].tobytes() + b"\n")
