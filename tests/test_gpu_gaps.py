"""Gap removal on the GPU (skh_remove_gaps_dev) against the oracle's RemoveGaps, record by record,
on the length distributions of the parity tests, with and without qualities, several gap sets."""
import ctypes as C

import numpy as np
import pytest

from helpers import make_batch, ptr, to_dev
from test_gpu_parity import CASES

pytestmark = pytest.mark.gpu


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


def oracle_remove(oracle_lib, seq, qual, offs, letters):
    oracle_lib.ora_remove_gaps.argtypes = [C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.c_size_t, C.c_char_p]
    oracle_lib.ora_remove_gaps.restype = C.c_size_t
    out_s, out_q, new_offs = [], [], [0]
    for r in range(len(offs) - 1):
        a, b = int(offs[r]), int(offs[r + 1])
        s = seq[a:b].copy()
        q = qual[a:b].copy() if qual is not None else None
        n = oracle_lib.ora_remove_gaps(s.ctypes.data_as(C.POINTER(C.c_uint8)),
                                       q.ctypes.data_as(C.POINTER(C.c_uint8)) if q is not None else None, b - a, letters)
        out_s.append(s[:n])
        if q is not None:
            out_q.append(q[:n])
        new_offs.append(new_offs[-1] + n)
    cat = lambda parts: np.concatenate(parts) if parts else np.zeros(0, np.uint8)
    return cat(out_s), (cat(out_q) if qual is not None else None), np.array(new_offs, np.uint64)


def run_gpu(lib, ctx, seq, qual, offs, letters):
    import torch
    nrec, total = len(offs) - 1, int(offs[-1])
    d_seq, d_offs = to_dev(seq), to_dev(offs)
    d_qual = to_dev(qual) if qual is not None else None
    d_out = torch.full((len(seq),), 0xEE, dtype=torch.uint8, device="cuda:0")
    d_qout = torch.full((len(seq),), 0xEE, dtype=torch.uint8, device="cuda:0") if qual is not None else None
    d_noffs = torch.full((nrec + 1,), -1, dtype=torch.int64, device="cuda:0")
    wsb = lib.skh_remove_gaps_workspace_size(nrec, total)
    d_ws = torch.zeros(wsb + 256, dtype=torch.uint8, device="cuda:0")
    ws = (d_ws.data_ptr() + 255) & ~255
    d_res = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    is_gap = (C.c_uint8 * 256)()
    for ch in letters:
        is_gap[ch] = 1
    torch.cuda.synchronize()
    lib.check(lib.skh_remove_gaps_dev(ctx, ptr(d_seq), ptr(d_qual) if qual is not None else None, ptr(d_offs), nrec, total,
                                      is_gap, ptr(d_out), ptr(d_qout) if qual is not None else None, ptr(d_noffs),
                                      C.c_void_p(ws), wsb, ptr(d_res), None))
    lib.check(lib.skh_sync(ctx))
    kept = int(d_res.cpu()[0])
    out = d_out.cpu().numpy()
    assert (out[kept:] == 0xEE).all()                 # nothing is written beyond the kept bytes
    return out[:kept], (d_qout.cpu().numpy()[:kept] if qual is not None else None), d_noffs.cpu().numpy().view(np.uint64)


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("letters", [b"- .", b"-", b"Nn-", b"ACGT"], ids=["default", "dash", "n_dash", "acgt"])
@pytest.mark.parametrize("with_qual", [False, True], ids=["fasta", "fastq"])
def test_remove_gaps_matches_oracle(lib, ctx, oracle_lib, case, letters, with_qual):
    seq, qual, offs = make_batch(CASES[case], seed=len(case) + len(letters), letters=b"ACGTacgtNn-. -.-", with_qual=with_qual)
    want_s, want_q, want_o = oracle_remove(oracle_lib, seq, qual, offs, letters)
    got_s, got_q, got_o = run_gpu(lib, ctx, seq, qual, offs, letters)
    assert np.array_equal(got_o, want_o)
    assert np.array_equal(got_s, want_s)
    if with_qual:
        assert np.array_equal(got_q, want_q)


def test_remove_gaps_nothing_and_everything(lib, ctx, oracle_lib):
    seq, qual, offs = make_batch([100, 0, 33, 5000], seed=3, letters=b"ACGT", with_qual=True)
    got_s, got_q, got_o = run_gpu(lib, ctx, seq, qual, offs, b"-")         # no gap byte anywhere
    total = int(offs[-1])
    assert np.array_equal(got_s, seq[:total]) and np.array_equal(got_q, qual[:total]) and np.array_equal(got_o, offs)
    got_s, got_q, got_o = run_gpu(lib, ctx, seq, qual, offs, b"ACGT")      # every byte is a gap
    assert len(got_s) == 0 and (got_o == 0).all()


def test_remove_gaps_at_bench_size(lib, ctx):
    # 2 M x 150 bp with 2 % gaps: kept bytes and offsets against torch boolean indexing
    import torch
    nrec, L = 2_000_000, 150
    g = torch.Generator(device="cuda:0"); g.manual_seed(5)
    letters = torch.tensor(list(b"ACGT" * 12 + b"-."), dtype=torch.uint8, device="cuda:0")
    seq = letters[torch.randint(0, len(letters), (nrec * L,), generator=g, device="cuda:0")]
    offs = torch.arange(nrec + 1, dtype=torch.int64, device="cuda:0") * L
    out = torch.empty_like(seq)
    noffs = torch.empty(nrec + 1, dtype=torch.int64, device="cuda:0")
    wsb = lib.skh_remove_gaps_workspace_size(nrec, nrec * L)
    d_ws = torch.zeros(wsb + 256, dtype=torch.uint8, device="cuda:0")
    res = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    is_gap = (C.c_uint8 * 256)()
    for ch in b"- .":
        is_gap[ch] = 1
    torch.cuda.synchronize()
    lib.check(lib.skh_remove_gaps_dev(ctx, ptr(seq), None, ptr(offs), nrec, nrec * L, is_gap, ptr(out), None, ptr(noffs),
                                      C.c_void_p((d_ws.data_ptr() + 255) & ~255), wsb, ptr(res), None))
    lib.check(lib.skh_sync(ctx))
    keep = (seq != ord("-")) & (seq != ord("."))
    assert int(res[0]) == int(keep.sum())
    assert torch.equal(out[:int(res[0])], seq[keep])
    want_offs = torch.cat([torch.zeros(1, dtype=torch.int64, device="cuda:0"), keep.view(nrec, L).sum(1).cumsum(0)])
    assert torch.equal(noffs, want_offs)


# ---------------------------------------------------------------- letters present per record (fx2tab -a)

def letters_of(seq, offs):
    out = np.zeros((len(offs) - 1, 8), np.uint32)
    for r in range(len(offs) - 1):
        for b in np.unique(seq[int(offs[r]):int(offs[r + 1])]):
            out[r, int(b) >> 5] |= np.uint32(1 << (int(b) & 31))
    return out


@pytest.mark.parametrize("case", sorted(CASES))
def test_letters_present_matches_numpy(lib, ctx, case):
    import torch
    seq, _, offs = make_batch(CASES[case], seed=len(case), letters=bytes(range(1, 256)))
    nrec, total = len(offs) - 1, int(offs[-1])
    d_seq, d_offs = to_dev(seq), to_dev(offs)
    d_p = torch.full((nrec * 8,), -1, dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    lib.check(lib.skh_letters_dev(ctx, ptr(d_seq), ptr(d_offs), nrec, total, ptr(d_p), None))
    lib.check(lib.skh_sync(ctx))
    got = d_p.cpu().numpy().view(np.uint32).reshape(nrec, 8)
    assert np.array_equal(got, letters_of(seq, offs))
