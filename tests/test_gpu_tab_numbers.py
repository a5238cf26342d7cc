"""fx2tab's numeric columns for EVERY (count, length) pair with length <= 2048: the device formatter's text
against the oracle's (C's "%.2f" on the same float64 arithmetic as fx2tab.go:127-143 -- two divisions rounded on
their own, then the sum; 0/0 spelled NaN).  The records exist only as offsets and counts: with --only-name no
sequence byte is read."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MAXLEN = 2048


def check_rows(oracle_lib, lens, g, c):
    """The device's rows for records of lengths `lens` with g G's and c C's against the oracle's."""
    import torch
    import seqkit_amd
    from seqkit_amd import capi
    lib = seqkit_amd.load()
    ctx = lib.init(0)
    dev = torch.device("cuda", 0)
    try:
        nrec = len(lens)
        offs = np.zeros(nrec + 1, np.uint64)
        np.cumsum(lens, out=offs[1:])
        counts = np.zeros((nrec, 4), np.uint64)
        counts[:, 0] = g
        counts[:, 1] = c

        oracle_lib.ora_tab_number_rows.restype = C.c_size_t
        oracle_lib.ora_tab_number_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        want = np.zeros(64 * nrec, np.uint8)
        nwant = oracle_lib.ora_tab_number_rows(g.ctypes.data, c.ctypes.data, lens.ctypes.data, nrec, want.ctypes.data)
        # every row of the device starts with the name "r" and the two empty fields of --only-name
        rows = want[:nwant].tobytes().split(b"\n")[:-1]
        assert len(rows) == nrec
        want_text = b"".join(b"r\t\t" + row + b"\n" for row in rows)

        text = np.zeros(16, np.uint8)
        text[0] = ord("r")
        d_text = torch.from_numpy(text).to(dev)
        d_npos = torch.zeros(nrec, dtype=torch.int64, device=dev)
        d_nlen = torch.ones(nrec, dtype=torch.int32, device=dev)
        d_offs = torch.from_numpy(offs.view(np.int64)).to(dev)
        d_counts = torch.from_numpy(counts.view(np.int64).reshape(-1)).to(dev)
        d_plane = torch.zeros(16, dtype=torch.uint8, device=dev)      # (never read)
        cap = len(want_text) + 64
        d_out = torch.zeros(cap, dtype=torch.uint8, device=dev)
        wsb = lib.skh_format_workspace_size(nrec)
        d_ws = torch.zeros(wsb + 256, dtype=torch.uint8, device=dev)
        ws = (d_ws.data_ptr() + 255) & ~255
        d_res = torch.zeros(2, dtype=torch.int64, device=dev)
        tf = capi.TabFormat()
        tf.ncols = 4; tf.only_id = 0; tf.print_seq = 0; tf.print_qual = 0
        for i, (k, terms) in enumerate([(capi.TAB_LENGTH, []), (capi.TAB_GC, [0, 1]), (capi.TAB_GC_SKEW, [0, 1]),
                                        (capi.TAB_CONTENT, [0])]):
            tf.cols[i].kind = k
            tf.cols[i].nterm = len(terms)
            for j, t in enumerate(terms):
                tf.cols[i].term[j] = t
        cptr = (C.c_void_p * 4)(d_counts.data_ptr(), None, None, None)
        ncl = (C.c_uint32 * 4)(4, 0, 0, 0)
        torch.cuda.synchronize()
        lib.check(lib.skh_format_tab_dev(ctx, d_text.data_ptr(), d_npos.data_ptr(), d_nlen.data_ptr(), d_plane.data_ptr(),
                                         None, d_offs.data_ptr(), nrec, cptr, ncl, C.byref(tf), d_out.data_ptr(), cap,
                                         ws, wsb, d_res.data_ptr(), None))
        lib.check(lib.skh_sync(ctx))
        res = capi.FormatResult.from_buffer_copy(d_res.cpu().numpy().tobytes())
        assert res.status == 0 and res.out_bytes == len(want_text)
        got = d_out[:res.out_bytes].cpu().numpy()
        ref = np.frombuffer(want_text, np.uint8)
        if not np.array_equal(got, ref):
            i = int(np.nonzero(got != ref)[0][0])
            raise AssertionError("first difference at byte %d: want %r got %r" %
                                 (i, want_text[max(0, i - 40):i + 20], got[max(0, i - 40):i + 20].tobytes()))
        return want_text
    finally:
        lib.skh_destroy(ctx)


def test_every_count_and_length(oracle_lib):
    lens = np.repeat(np.arange(MAXLEN + 1, dtype=np.uint64), np.arange(1, MAXLEN + 2))
    first = np.cumsum(np.arange(1, MAXLEN + 2)) - np.arange(1, MAXLEN + 2)
    g = (np.arange(len(lens), dtype=np.uint64) - np.repeat(first, np.arange(1, MAXLEN + 2)).astype(np.uint64))
    assert g.max() == MAXLEN and bool((g <= lens).all())
    c = (lens - g) // np.uint64(3)                      # some C besides the G: the sum of two quotients
    want_text = check_rows(oracle_lib, lens, g, c)
    # the pairs the reference's documentation shows (doc/docs/usage.md: 41.87 etc. come from rounding twice)
    assert b"\t50.00\t" in want_text and b"NaN" in want_text


def test_long_records_and_ties(oracle_lib):
    """Percentages of records up to 65535 bases are printed from one integer division unless the hundredths are an
    exact tie; longer records and ties take the float64 arithmetic.  Every count of lengths either side of that
    limit and of lengths rich in ties (x.xx5 exactly: multiples of 2000, powers of two), and random pairs up to
    2^20 bases."""
    rng = np.random.default_rng(9)
    full = [65535, 65536, 65534, 64000, 60000, 40000, 32768, 20000, 16384, 8000, 6000, 4000, 2000, 65537, 70000]
    lens = np.concatenate([np.full(n + 1, n, np.uint64) for n in full])
    g = np.concatenate([np.arange(n + 1, dtype=np.uint64) for n in full])
    rl = rng.integers(1, 1 << 20, 400_000).astype(np.uint64)
    rg = (rng.random(400_000) * (rl + np.uint64(1))).astype(np.uint64)
    lens = np.concatenate([lens, rl]); g = np.concatenate([g, np.minimum(rg, rl)])
    c = (lens - g) // np.uint64(3)
    want_text = check_rows(oracle_lib, lens, g, c)
    assert want_text.count(b".") == 3 * len(lens)
