#!/usr/bin/env python3
"""Writes a synthetic FASTQ (or FASTA) text file of fixed-length reads, vectorised with numpy:
usage: gen_fastx.py out.fq NREC LEN [seed] [--fasta [--width W]]"""
import sys
import numpy as np

out, nrec, L = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
seed = int(sys.argv[4], 0) if len(sys.argv) > 4 and not sys.argv[4].startswith("--") else 0xC2
fasta = "--fasta" in sys.argv
width = int(sys.argv[sys.argv.index("--width") + 1]) if "--width" in sys.argv else 0
rng = np.random.default_rng(seed)
CH = 200_000
with open(out, "wb") as fh:
    for a in range(0, nrec, CH):
        n = min(CH, nrec - a)
        seq = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (n, L))]
        seq[rng.random((n, L)) < 2.5e-5] = ord("N")
        ids = np.arange(a, a + n)
        name = np.char.add(b"SIM:1:FC:1:", np.char.zfill(ids.astype("S"), 9))
        name = np.char.add(name, b" 1:N:0:ATCACG")
        if fasta:
            if width:
                nl = (L + width - 1) // width
                body = np.full((n, nl * (width + 1)), ord("\n"), np.uint8)
                for i in range(nl):
                    w = min(width, L - i * width)
                    body[:, i * (width + 1): i * (width + 1) + w] = seq[:, i * width: i * width + w]
                keep = np.ones(nl * (width + 1), bool)
                last_w = L - (nl - 1) * width
                keep[(nl - 1) * (width + 1) + last_w: nl * (width + 1) - 1] = False
                body = body[:, keep]
            else:
                body = np.concatenate([seq, np.full((n, 1), ord("\n"), np.uint8)], axis=1)
            hdr = np.char.add(np.char.add(b">", name), b"\n")
            W = hdr.dtype.itemsize
            rec = np.concatenate([np.frombuffer(hdr.tobytes(), np.uint8).reshape(n, W), body], axis=1)
        else:
            qual = rng.integers(ord("#"), ord("I") + 1, (n, L), dtype=np.uint8)
            hdr = np.char.add(np.char.add(b"@", name), b"\n")
            W = hdr.dtype.itemsize
            nlc = np.full((n, 1), ord("\n"), np.uint8)
            plus = np.tile(np.frombuffer(b"+\n", np.uint8), (n, 1))
            rec = np.concatenate([np.frombuffer(hdr.tobytes(), np.uint8).reshape(n, W), seq, nlc, plus, qual, nlc], axis=1)
        fh.write(rec.tobytes())
