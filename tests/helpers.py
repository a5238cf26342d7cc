"""Test-side helpers: numpy wrappers around the CPU oracle, synthetic CSR batches and
torch device buffers.  Nothing here is product code."""
import ctypes as C
import gzip
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

u8p, u64p, i64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.POINTER(C.c_int64)


def _p(a, t):
    return a.ctypes.data_as(t)


class Oracle:
    """numpy front end of oracle/libseqkit_oracle.so (the checker)."""

    def __init__(self, dll):
        self.dll = dll
        dll.ora_batch_transform.argtypes = [u8p, u8p, u64p, C.c_uint64, u8p, C.c_int]
        dll.ora_batch_transform.restype = None
        dll.ora_batch_count.argtypes = [u8p, u64p, C.c_uint64, u8p, C.c_uint32, u64p]
        dll.ora_batch_count.restype = None
        dll.ora_batch_stats.argtypes = [u8p, u64p, C.c_uint64, u8p, C.c_size_t, i64p]
        dll.ora_batch_stats.restype = None
        dll.ora_pair_table.argtypes = [C.c_int, u8p]
        dll.ora_pair_table.restype = None
        dll.ora_guess_alphabet.argtypes = [u8p, C.c_size_t, C.c_size_t]
        dll.ora_guess_alphabet.restype = C.c_int

    def pair_table(self, alphabet):
        t = np.zeros(256, np.uint8)
        self.dll.ora_pair_table(alphabet, _p(t, u8p))
        return t

    def transform(self, seq, qual, offs, lut, do_reverse):
        s = seq.copy()
        q = qual.copy() if qual is not None else None
        self.dll.ora_batch_transform(_p(s, u8p), _p(q, u8p) if q is not None else None, _p(offs, u64p),
                                     len(offs) - 1, _p(np.ascontiguousarray(lut, np.uint8), u8p)
                                     if lut is not None else None, int(do_reverse))
        return s, q

    def count(self, seq, offs, class_of, nclass):
        nrec = len(offs) - 1
        out = np.zeros(nrec * nclass, np.uint64)
        self.dll.ora_batch_count(_p(seq, u8p), _p(offs, u64p), nrec,
                                 _p(np.ascontiguousarray(class_of, np.uint8), u8p), nclass, _p(out, u64p))
        return out.reshape(nrec, nclass)

    def stats(self, seq, offs, gap_letters=b"- ."):
        out = np.zeros(5, np.int64)
        g = np.frombuffer(gap_letters, np.uint8).copy()
        self.dll.ora_batch_stats(_p(seq, u8p), _p(offs, u64p), len(offs) - 1, _p(g, u8p), len(g), _p(out, i64p))
        return tuple(int(x) for x in out)


# ---------------------------------------------------------------- synthetic batches

DNA_SOFT = (b"ACGTacgtNnRYKMSWBDHVrykmswbdhv-. *Uu", None)


def make_batch(lengths, seed, letters=b"ACGT", probs=None, with_qual=False):
    """CSR batch with iid bytes drawn from `letters`; the planes are padded to 16 bytes."""
    rng = np.random.default_rng(seed)
    lengths = np.asarray(lengths, np.uint64)
    offs = np.zeros(len(lengths) + 1, np.uint64)
    np.cumsum(lengths, out=offs[1:])
    total = int(offs[-1])
    padded = (total + 15) & ~15
    lt = np.frombuffer(letters, np.uint8)
    seq = np.zeros(max(padded, 16), np.uint8)
    seq[:total] = lt[rng.choice(len(lt), size=total, p=probs)] if total else seq[:0]
    qual = None
    if with_qual:
        qual = np.zeros(max(padded, 16), np.uint8)
        qual[:total] = rng.integers(33, 74, size=total, dtype=np.uint8)
    return seq, qual, offs


def read_fastx_plain(path):
    """Minimal FASTA/FASTQ loader for the fixtures (names, CSR seq, CSR qual)."""
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rb") as fh:
        data = fh.read()
    names, seqs, quals = [], [], []
    lines = data.split(b"\n")
    if data[:1] == b">":
        cur = None
        for l in lines:
            if l.startswith(b">"):
                names.append(l[1:]); cur = []; seqs.append(cur)
            elif cur is not None:
                cur.append(l.rstrip(b"\r"))
        seqs = [b"".join(s) for s in seqs]
    else:
        i = 0
        while i + 3 < len(lines) + 1 and i < len(lines) and lines[i].startswith(b"@"):
            names.append(lines[i][1:]); seqs.append(lines[i + 1]); quals.append(lines[i + 3]); i += 4
    lens = np.array([len(s) for s in seqs], np.uint64)
    offs = np.zeros(len(seqs) + 1, np.uint64)
    np.cumsum(lens, out=offs[1:])
    total = int(offs[-1]); padded = max((total + 15) & ~15, 16)
    seq = np.zeros(padded, np.uint8); seq[:total] = np.frombuffer(b"".join(seqs), np.uint8)
    qual = None
    if quals:
        qual = np.zeros(padded, np.uint8); qual[:total] = np.frombuffer(b"".join(quals), np.uint8)
    return names, seq, qual, offs


# ---------------------------------------------------------------- device side (torch = memory only)

def to_dev(a, device="cuda:0"):
    import torch
    if a.dtype == np.uint64:
        return torch.from_numpy(a.view(np.int64)).to(device)
    return torch.from_numpy(a).to(device)


def dev_empty_like_plane(n, device="cuda:0", fill=0xEE):
    import torch
    return torch.full((n,), fill, dtype=torch.uint8, device=device)


def ptr(t):
    return C.c_void_p(t.data_ptr())


# ---------------------------------------------------------------- text fixtures

def wrapped_fastq(path, width=60, four_line_head=0, crlf=False):
    """reads_1.fq.gz with sequence and quality wrapped at `width` columns (quality lines then start with
    '@' and '+' now and then); the first `four_line_head` records stay four-line."""
    import gzip
    with gzip.open(os.path.join(GOLDEN, "reads_1.fq.gz"), "rb") as fh:
        lines = fh.read().split(b"\n")
    eol = b"\r\n" if crlf else b"\n"
    out = []
    for i in range(0, len(lines) - 3, 4):
        name, seq, qual = lines[i], lines[i + 1], lines[i + 3]
        w = len(seq) + 1 if i // 4 < four_line_head else width
        out.append(name)
        out += [seq[k:k + w] for k in range(0, len(seq), w)] or [b""]
        out.append(b"+" + (name[1:] if i % 8 == 0 else b""))
        out += [qual[k:k + w] for k in range(0, len(qual), w)] or [b""]
        if i % 40 == 0:
            out.append(b"")                           # blank lines between records
    with open(path, "wb") as fh:
        fh.write(eol.join(out) + eol)
