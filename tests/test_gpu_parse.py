"""GPU parity of the text -> batch parser (skh_parse_dev) against the oracle's reader
(oracle/seqkit_oracle.c, ora_reader_*), which is pinned by the reference's own vectors
(names/IDs, hairpin record lengths, the seq -w 60 identity round trip).  Bit-exact: names,
sequence plane, quality plane and offsets; block-wise parsing (non-final blocks + carry) must
give the same records as one final block."""
import ctypes as C
import gzip
import os
import zlib

import numpy as np
import pytest

from helpers import Oracle

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class OraRecord(C.Structure):
    _fields_ = [("name", C.POINTER(C.c_uint8)), ("name_len", C.c_size_t), ("id_len", C.c_size_t),
                ("id", C.POINTER(C.c_uint8)), ("seq", C.POINTER(C.c_uint8)), ("seq_len", C.c_size_t),
                ("qual", C.POINTER(C.c_uint8)), ("qual_len", C.c_size_t)]


def oracle_records(ora, path):
    """[(name, seq, qual)] as the oracle's reader sees the file; raises ValueError on a parse error."""
    dll = ora.dll
    dll.ora_reader_open.restype = C.c_void_p
    dll.ora_reader_open.argtypes = [C.c_char_p, C.c_int, C.c_size_t]
    dll.ora_reader_next.restype = C.c_int
    dll.ora_reader_next.argtypes = [C.c_void_p, C.POINTER(OraRecord)]
    dll.ora_reader_close.argtypes = [C.c_void_p]
    r = dll.ora_reader_open(path.encode(), -1, 10000)
    assert r
    out, rec = [], OraRecord()
    try:
        while True:
            rc = dll.ora_reader_next(r, C.byref(rec))
            if rc == 0:
                break
            if rc < 0:
                raise ValueError("oracle: parse error")
            out.append((C.string_at(rec.name, rec.name_len), C.string_at(rec.seq, rec.seq_len),
                        C.string_at(rec.qual, rec.qual_len)))
    finally:
        dll.ora_reader_close(r)
    return out


@pytest.fixture(scope="module")
def env(oracle_lib):
    import torch
    import seqkit_amd
    from seqkit_amd import capi
    lib = seqkit_amd.load()
    ctx = lib.init(0)
    yield {"lib": lib, "ctx": ctx, "torch": torch, "capi": capi, "ora": Oracle(oracle_lib)}
    lib.skh_destroy(ctx)


def gpu_parse(env, text, fmt, final, max_lines=None, max_records=None, seq_capacity=None):
    """One skh_parse_dev call on `text` (bytes).  Returns (result, names, seq, qual, offsets)."""
    torch, lib, capi = env["torch"], env["lib"], env["capi"]
    dev = torch.device("cuda", 0)
    n = len(text)
    padded = max((n + 15) & ~15, 16)
    h = np.zeros(padded, np.uint8)
    h[:n] = np.frombuffer(text, np.uint8)
    d_text = torch.from_numpy(h).to(dev)
    if max_lines is None:
        max_lines = text.count(b"\n") + 2
    if max_records is None:
        max_records = max_lines
    if seq_capacity is None:
        seq_capacity = padded
    d_seq = torch.full((max(seq_capacity, 16),), 0xEE, dtype=torch.uint8, device=dev)
    d_qual = torch.full((max(seq_capacity, 16),), 0xEE, dtype=torch.uint8, device=dev)
    d_offs = torch.full((max_records + 1,), -1, dtype=torch.int64, device=dev)
    d_npos = torch.zeros(max(max_records, 1), dtype=torch.int64, device=dev)
    d_nlen = torch.zeros(max(max_records, 1), dtype=torch.int32, device=dev)
    ws_bytes = (lib.skh_parse_wrapped_workspace_size(n, max_lines, max_records) if fmt == capi.FASTQ_WRAPPED
                else lib.skh_parse_workspace_size(n, max_lines))
    d_ws = torch.zeros(ws_bytes + 256, dtype=torch.uint8, device=dev)
    ws_ptr = (d_ws.data_ptr() + 255) & ~255
    d_res = torch.zeros(6, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    lib.check(lib.skh_parse_dev(env["ctx"], d_text.data_ptr(), n, fmt, int(final), d_seq.data_ptr(), d_qual.data_ptr(),
                                seq_capacity, d_offs.data_ptr(), d_npos.data_ptr(), d_nlen.data_ptr(), max_records,
                                ws_ptr, ws_bytes, max_lines, d_res.data_ptr(), None))
    lib.check(lib.skh_sync(env["ctx"]))
    res = capi.ParseResult.from_buffer_copy(d_res.cpu().numpy().tobytes())
    if res.status != capi.PARSE_OK:
        return res, None, None, None, None
    nrec, sb = res.nrec, res.seq_bytes
    offs = d_offs[:nrec + 1].cpu().numpy().astype(np.uint64)
    npos = d_npos[:nrec].cpu().numpy()
    nlen = d_nlen[:nrec].cpu().numpy()
    names = [text[int(p):int(p) + int(l)] for p, l in zip(npos, nlen)]
    seq = d_seq[:sb].cpu().numpy().tobytes()
    qual = d_qual[:sb].cpu().numpy().tobytes() if fmt != capi.FASTA else None
    # nothing may be written past the sequence bytes
    assert bool((d_seq[sb:] == 0xEE).all())
    return res, names, seq, qual, offs


def records_of(names, seq, qual, offs):
    out = []
    for r, nm in enumerate(names):
        a, b = int(offs[r]), int(offs[r + 1])
        out.append((nm, seq[a:b], qual[a:b] if qual is not None else b""))
    return out


def check_text(env, tmp_path, text, fmt, tag):
    path = str(tmp_path / ("in_%s.%s" % (tag, "fa" if fmt == env["capi"].FASTA else "fq")))
    with open(path, "wb") as fh:
        fh.write(text)
    want = oracle_records(env["ora"], path)
    res, names, seq, qual, offs = gpu_parse(env, text, fmt, True)
    assert res.status == env["capi"].PARSE_OK, (res.status, res.error_line)
    assert res.consumed == len(text)
    got = records_of(names, seq, qual, offs)
    assert len(got) == len(want)
    assert got == want
    return want


# ---------------------------------------------------------------- fixtures of the reference

@pytest.mark.parametrize("name", ["hairpin.fa.gz", "mature.fa.gz", "contigs.fa", "Illimina1.8.fq.gz", "reads_1.fq.gz",
                                  "Illimina1.5.fq"])
def test_fixture_files(env, tmp_path, name):
    p = os.path.join(GOLD, name)
    data = gzip.open(p, "rb").read() if p.endswith(".gz") else open(p, "rb").read()
    fmt = env["capi"].FASTQ if data[:1] == b"@" else env["capi"].FASTA
    want = check_text(env, tmp_path, data, fmt, name.split(".")[0])
    assert len(want) > 0


# ---------------------------------------------------------------- synthetic edge cases

FASTA_CASES = {
    "empty": b"",
    "one_no_newline": b">a\nACGT",
    "one_newline": b">a\nACGT\n",
    "crlf": b">a desc\r\nAC\r\nGT\r\n>b\r\nTTTT\r\n",
    "blank_lines": b"\n\n>a\n\nAC\n\nGT\n\n>b\nA\n\n",
    "empty_records": b">a\n>b\n>c\nA\n>d\n",
    "empty_name": b">\nACGT\n> \nAC\n",
    "gt_inside": b">a\nAC>GT\nA>\n",
    "spaces_kept": b">s p a c e\nAC GT\n A \n",
    "long_lines": b">x\n" + b"ACGT" * 5000 + b"\n" + b"TTGA" * 3000 + b"\n>y\n" + b"G" * 70001 + b"\n",
    "only_head": b">a",
    "tabs": b">id\tdesc\nAC\tGT\n",
}


@pytest.mark.parametrize("case", sorted(FASTA_CASES))
def test_fasta_cases(env, tmp_path, case):
    check_text(env, tmp_path, FASTA_CASES[case], env["capi"].FASTA, case)


FASTQ_CASES = {
    "empty": b"",
    "one": b"@r1\nACGT\n+\nIIII\n",
    "one_no_newline": b"@r1\nACGT\n+\nIIII",
    "plus_name": b"@r1 x\nACGT\n+r1 x\nIIII\n@r2\nAC\n+\n#I\n",
    "crlf": b"@r1\r\nACGT\r\n+\r\nIIII\r\n",
    "at_in_quality": b"@r1\nACGT\n+\n@III\n@r2\nAC\n+\n@@\n",
    "plus_in_quality": b"@r1\nACGT\n+\n+III\n",
    "empty_seq": b"@r1\n\n+\n\n@r2\nA\n+\nI\n",
    "trailing_blank": b"@r1\nACGT\n+\nIIII\n\n\n",
    "long": b"@L\n" + b"ACGT" * 20000 + b"\n+\n" + b"I#" * 40000 + b"\n",
}


@pytest.mark.parametrize("case", sorted(FASTQ_CASES))
def test_fastq_cases(env, tmp_path, case):
    check_text(env, tmp_path, FASTQ_CASES[case], env["capi"].FASTQ, case)


BAD = {
    "fa_text_before_head": (b"ACGT\n>a\nAC\n", 1, 0),
    "fq_bad_head": (b"@r1\nAC\n+\nII\nr2\nAC\n+\nII\n", 2, 4),
    "fq_bad_plus": (b"@r1\nAC\nII\nII\n", 2, 2),
    "fq_len_mismatch": (b"@r1\nACG\n+\nII\n", 2, 3),
    "fq_truncated": (b"@r1\nAC\n+\nII\n@r2\nAC\n", 2, 4),
    "fq_multiline": (b"@r1\nAC\nGT\n+\nIIII\n", 2, 2),
    "fq_blank_between": (b"@r1\nAC\n+\nII\n\n@r2\nAC\n+\nII\n", 2, 4),
}


@pytest.mark.parametrize("case", sorted(BAD))
def test_format_errors(env, case):
    text, fmt, line = BAD[case]
    res, *_ = gpu_parse(env, text, fmt, True)
    assert res.status == env["capi"].PARSE_EFORMAT
    assert res.error_line == line


def test_capacity_errors(env):
    capi = env["capi"]
    text = b">a\nACGT\n>b\nAC\n>c\nA\n"
    assert gpu_parse(env, text, capi.FASTA, True, max_lines=3)[0].status == capi.PARSE_ECAPACITY
    assert gpu_parse(env, text, capi.FASTA, True, max_records=2)[0].status == capi.PARSE_ECAPACITY
    assert gpu_parse(env, text, capi.FASTA, True, seq_capacity=5)[0].status == capi.PARSE_ECAPACITY
    assert gpu_parse(env, text, capi.FASTA, True, max_records=3, seq_capacity=7)[0].status == capi.PARSE_OK


# ---------------------------------------------------------------- random texts, whole and block-wise

def random_fasta(rng, nrec, crlf):
    eol = b"\r\n" if crlf else b"\n"
    parts = []
    for r in range(nrec):
        parts.append(b">rec%d some description %d" % (r, rng.integers(0, 1 << 30)) + eol)
        L = int(rng.choice([0, 1, 15, 16, 17, 59, 60, 61, 150, 1000, int(rng.integers(0, 5000))]))
        s = np.frombuffer(b"ACGTNacgtn-RYKM", np.uint8)[rng.integers(0, 15, L)].tobytes()
        w = int(rng.choice([60, 70, 80, 1 << 30]))
        for a in range(0, L, w):
            parts.append(s[a:a + w] + eol)
        if rng.random() < 0.05:
            parts.append(eol)
    return b"".join(parts)


def random_fastq(rng, nrec, crlf):
    eol = b"\r\n" if crlf else b"\n"
    parts = []
    for r in range(nrec):
        L = int(rng.choice([0, 1, 16, 100, 150, 151, 250, int(rng.integers(0, 3000))]))
        s = np.frombuffer(b"ACGTN", np.uint8)[rng.integers(0, 5, L)].tobytes()
        q = rng.integers(33, 75, L, dtype=np.uint8).tobytes()     # includes '@' (64) and '+' (43)
        parts += [b"@read%d 1:N:0" % r + eol, s + eol, (b"+" if r % 3 else b"+read%d 1:N:0" % r) + eol, q + eol]
    return b"".join(parts)


@pytest.mark.parametrize("kind,crlf", [("fa", False), ("fa", True), ("fq", False), ("fq", True)])
def test_random_whole_and_blockwise(env, tmp_path, kind, crlf):
    capi = env["capi"]
    rng = np.random.default_rng(zlib.crc32(("%s%d" % (kind, crlf)).encode()))
    text = random_fasta(rng, 3000, crlf) if kind == "fa" else random_fastq(rng, 3000, crlf)
    fmt = capi.FASTA if kind == "fa" else capi.FASTQ
    want = check_text(env, tmp_path, text, fmt, "rnd_%s_%d" % (kind, crlf))
    # block-wise: fixed-size reads with the unconsumed tail carried over, as a driver would do
    for block in (4096, 65536, 1 << 20):
        got, pos, carry = [], 0, b""
        while True:
            chunk = text[pos:pos + block]
            pos += len(chunk)
            final = pos >= len(text)
            buf = carry + chunk
            res, names, seq, qual, offs = gpu_parse(env, buf, fmt, final)
            assert res.status == capi.PARSE_OK, (block, res.status, res.error_line)
            got += records_of(names, seq, qual, offs)
            carry = buf[res.consumed:]
            if final:
                assert res.consumed == len(buf)
                break
            if res.nrec == 0 and len(chunk) == 0:
                break
        assert got == want, block


def test_large_fastq_properties(env):
    """4-line FASTQ at a size the oracle would take long on: structure checks only."""
    capi = env["capi"]
    rng = np.random.default_rng(5)
    nrec, L = 200_000, 150
    seq = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (nrec, L))]
    qual = rng.integers(35, 74, (nrec, L), dtype=np.uint8)
    nl = np.full((nrec, 1), 10, np.uint8)
    head = np.frombuffer((b"@SIM:%09d\n" * 1) % 0, np.uint8)
    heads = np.tile(head, (nrec, 1))
    ids = np.char.zfill(np.arange(nrec).astype("S"), 9)
    heads[:, 5:14] = np.frombuffer(ids.tobytes(), np.uint8).reshape(nrec, 9)
    plus = np.tile(np.frombuffer(b"+\n", np.uint8), (nrec, 1))
    text = np.concatenate([heads, seq, nl, plus, qual, nl], axis=1).tobytes()
    res, names, s, q, offs = gpu_parse(env, text, capi.FASTQ, True)
    assert res.status == capi.PARSE_OK and res.nrec == nrec and res.seq_bytes == nrec * L
    assert np.array_equal(offs, np.arange(nrec + 1, dtype=np.uint64) * L)
    assert s == seq.tobytes() and q == qual.tobytes()
    assert names[0] == b"SIM:000000000" and names[-1] == b"SIM:%09d" % (nrec - 1)


# ---------------------------------------------------------------- FASTQ with wrapped records

def wrapped_fastq(rng, nrec, crlf, width=None, blank=0.1):
    """Records whose sequence and quality are wrapped (each at its own width), quality lines that start with
    '@' and '+', blank lines between records, records without sequence, four-line records among them."""
    eol = b"\r\n" if crlf else b"\n"
    parts = []
    for r in range(nrec):
        L = int(rng.choice([0, 1, 7, 60, 61, 150, 151, 400, int(rng.integers(0, 3000))]))
        s = np.frombuffer(b"ACGTN", np.uint8)[rng.integers(0, 5, L)].tobytes()
        q = rng.integers(33, 75, L, dtype=np.uint8)
        if L and rng.random() < 0.5:
            q[rng.integers(0, L, max(1, L // 20))] = rng.choice([ord("@"), ord("+")])
        q = q.tobytes()
        ws = width or int(rng.choice([1, 13, 60, 70, 1 << 30]))
        wq = width or int(rng.choice([1, 17, 60, 70, 1 << 30]))
        parts.append(b"@read%d some text" % r + eol)
        for a in range(0, L, ws):
            parts.append(s[a:a + ws] + eol)
        parts.append((b"+" if r % 3 else b"+read%d" % r) + eol)
        for a in range(0, L, wq):
            parts.append(q[a:a + wq] + eol)
        if rng.random() < blank:
            parts.append(eol * int(rng.integers(1, 3)))
    return b"".join(parts)


@pytest.mark.parametrize("crlf", [False, True])
def test_wrapped_fastq_whole_and_blockwise(env, tmp_path, crlf):
    capi = env["capi"]
    rng = np.random.default_rng(40 + crlf)
    text = wrapped_fastq(rng, 2500, crlf)
    path = str(tmp_path / "w.fq")
    open(path, "wb").write(text)
    want = oracle_records(env["ora"], path)
    assert len(want) == 2500
    res, names, seq, qual, offs = gpu_parse(env, text, capi.FASTQ_WRAPPED, True)
    assert res.status == capi.PARSE_OK and res.wrapped == 1 and res.consumed == len(text), (res.status, res.consumed)
    assert records_of(names, seq, qual, offs) == want
    for block in (3000, 50_000, 1 << 20):
        got, pos, carry = [], 0, b""
        while True:
            chunk = text[pos:pos + block]
            pos += len(chunk)
            final = pos >= len(text)
            buf = carry + chunk
            res, names, seq, qual, offs = gpu_parse(env, buf, capi.FASTQ_WRAPPED, final)
            assert res.status == capi.PARSE_OK, (block, pos, res.status)
            got += records_of(names, seq, qual, offs)
            carry = buf[res.consumed:]
            if final:
                assert res.consumed == len(buf)
                break
            assert len(carry) < 4 * 3100 * 2 + 3 * block, "no progress"
        assert got == want, block


def test_wrapped_fastq_four_line_text_and_errors(env, tmp_path):
    capi = env["capi"]
    rng = np.random.default_rng(5)
    text = random_fastq(rng, 1000, False)
    path = str(tmp_path / "p.fq")
    open(path, "wb").write(text)
    res, names, seq, qual, offs = gpu_parse(env, text, capi.FASTQ_WRAPPED, True)
    assert res.status == capi.PARSE_OK          # (wrapped = 1: it holds records without bases, whose quality line is blank)
    assert records_of(names, seq, qual, offs) == oracle_records(env["ora"], path)
    plain = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, b"ACGT" * (1 + i % 40), b"I@+J" * (1 + i % 40)) for i in range(3000))
    res = gpu_parse(env, plain, capi.FASTQ_WRAPPED, True)[0]
    assert res.status == capi.PARSE_OK and res.nrec == 3000 and res.wrapped == 0
    for bad in (b"@r\nACGT\nAC\n+\nIIII\nIII\n",        # more quality than sequence
                b"ACGT\n@r\nAC\n+\nII\n",                  # text before the first record
                b"@r\nACGT\n+\nII\n",                      # the file ends inside a record
                b"@r\nACGT\n+\nIIII\nxx\n"):               # a line that is no '@' line where a record must start
        res = gpu_parse(env, bad, capi.FASTQ_WRAPPED, True)[0]
        assert res.status == capi.PARSE_EFORMAT and res.error_line == -1, bad
    # not final: the record at the end is left for the next block
    res, names, seq, qual, offs = gpu_parse(env, b"@a\nAC\nGT\n+\nII\nII\n@b\nACGT\n+\nII\n", capi.FASTQ_WRAPPED, False)
    assert res.status == capi.PARSE_OK and res.nrec == 1 and res.consumed == 17 and seq == b"ACGT" and qual == b"IIII"
    res = gpu_parse(env, b"", capi.FASTQ_WRAPPED, True)[0]
    assert res.status == capi.PARSE_OK and res.nrec == 0


def test_wrapped_fastq_capacity_errors(env):
    capi = env["capi"]
    text = b"@a\nAC\nGT\n+\nII\nII\n@b\nACG\n+\nIII\n@c\nA\n+\nI\n"
    W = capi.FASTQ_WRAPPED
    assert gpu_parse(env, text, W, True)[0].nrec == 3
    assert gpu_parse(env, text, W, True, max_lines=5)[0].status == capi.PARSE_ECAPACITY
    assert gpu_parse(env, text, W, True, max_records=2)[0].status == capi.PARSE_ECAPACITY
    assert gpu_parse(env, text, W, True, seq_capacity=7)[0].status == capi.PARSE_ECAPACITY
    res, names, seq, qual, offs = gpu_parse(env, text, W, True, max_records=3, seq_capacity=8)
    assert res.status == capi.PARSE_OK and seq == b"ACGTACGA" and qual == b"IIIIIIII" and names == [b"a", b"b", b"c"]


def test_wrapped_fastq_damaged_texts(env, tmp_path):
    """Small wrapped texts with a line deleted, doubled, inserted or re-headed, or cut at any byte: the device parser
    accepts exactly the texts the reader restated from the reference accepts, with the same records."""
    capi = env["capi"]
    rng = np.random.default_rng(77)
    path = str(tmp_path / "d.fq")
    n_ok = n_bad = 0
    for case in range(1400):
        crlf = bool(case & 1)
        eol = b"\r\n" if crlf else b"\n"
        text = wrapped_fastq(rng, int(rng.integers(1, 6)), crlf, blank=0.3)
        lines = text.split(eol)
        tail = lines.pop()                                   # b"" behind the last line end
        assert tail == b""
        kind = case % 7
        k = int(rng.integers(0, len(lines)))
        if kind == 0:
            del lines[k]
        elif kind == 1:
            lines.insert(k, lines[k])
        elif kind == 2:
            lines.insert(k, [b"@x y", b"+", b"ACGT", b"", b"+@+@"][int(rng.integers(0, 5))])
        elif kind == 3 and lines[k]:
            lines[k] = [b"@", b"+", b"A"][int(rng.integers(0, 3))] + lines[k][1:]
        text = eol.join(lines) + eol
        if kind == 4:
            text = text[:int(rng.integers(0, len(text) + 1))]
        elif kind == 5:
            text = text[:-len(eol)]                          # no line end behind the last line
        open(path, "wb").write(text)
        try:
            want = oracle_records(env["ora"], path)
        except ValueError:
            want = None
        res, names, seq, qual, offs = gpu_parse(env, text, capi.FASTQ_WRAPPED, True)
        if want is None:
            n_bad += 1
            assert res.status == capi.PARSE_EFORMAT, (case, kind, text)
        else:
            n_ok += 1
            assert res.status == capi.PARSE_OK and res.consumed == len(text), (case, kind, res.status, text)
            assert records_of(names, seq, qual, offs) == want, (case, kind, text)
    assert n_ok > 50 and n_bad > 50, (n_ok, n_bad)


@pytest.mark.parametrize("fmt_name", ["fasta", "fastq"])
def test_wide_position_tables(env, tmp_path, fmt_name, monkeypatch):
    """A text of 4 GiB or more keeps its tables of positions in 64 bits; SKH_PARSE_POS64 asks for that form on a small
    text: same records, whole and block by block."""
    capi = env["capi"]
    rng = np.random.default_rng(11)
    fmt = capi.FASTA if fmt_name == "fasta" else capi.FASTQ
    text = random_fasta(rng, 3000, False) if fmt_name == "fasta" else random_fastq(rng, 3000, False)
    narrow = gpu_parse(env, text, fmt, True)
    monkeypatch.setenv("SKH_PARSE_POS64", "1")
    wide = gpu_parse(env, text, fmt, True)
    assert narrow[0].status == wide[0].status == capi.PARSE_OK and narrow[0].nrec == wide[0].nrec
    assert records_of(*narrow[1:]) == records_of(*wide[1:])
    check_text(env, tmp_path, text, fmt, "wide_" + fmt_name)
    cut = len(text) // 2
    res = gpu_parse(env, text[:cut], fmt, False)[0]
    monkeypatch.delenv("SKH_PARSE_POS64")
    res_narrow = gpu_parse(env, text[:cut], fmt, False)[0]
    assert (res.status, res.nrec, res.consumed, res.seq_bytes) == (res_narrow.status, res_narrow.nrec, res_narrow.consumed, res_narrow.seq_bytes)
