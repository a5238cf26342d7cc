"""GPU parity of the whole text path of `seq`: text -> skh_parse_dev -> (skh_revcomp_dev) ->
skh_format_seq_dev -> text, against the bytes the oracle's `seq` command writes for the same
file and flags (oracle/oracle_cli.c, pinned by the reference's documented outputs and by the
`seq -w 60` identity round trip on hairpin.fa).  Bit-exact."""
import ctypes as C
import gzip
import os
import subprocess
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def env(oracle_cli, oracle_lib):
    import torch
    import seqkit_amd
    from seqkit_amd import capi
    lib = seqkit_amd.load()
    ctx = lib.init(0)
    from helpers import Oracle
    yield {"lib": lib, "ctx": ctx, "torch": torch, "capi": capi, "cli": oracle_cli, "ora": Oracle(oracle_lib)}
    lib.skh_destroy(ctx)


def aligned_ws(torch, nbytes, dev):
    t = torch.zeros(nbytes + 256, dtype=torch.uint8, device=dev)
    return t, (t.data_ptr() + 255) & ~255


def gpu_seq_command(env, text, flags):
    """What `seq <flags>` prints for `text`, computed on the GPU end to end."""
    torch, lib, capi, ctx = env["torch"], env["lib"], env["capi"], env["ctx"]
    dev = torch.device("cuda", 0)
    n = len(text)
    fastq = text[:1] == b"@"
    padded = max((n + 15) & ~15, 16)
    h = np.zeros(padded, np.uint8); h[:n] = np.frombuffer(text, np.uint8)
    d_text = torch.from_numpy(h).to(dev)
    max_lines = text.count(b"\n") + 2
    d_seq = torch.zeros(padded, dtype=torch.uint8, device=dev)
    d_qual = torch.zeros(padded, dtype=torch.uint8, device=dev)
    d_offs = torch.zeros(max_lines + 1, dtype=torch.int64, device=dev)
    d_npos = torch.zeros(max_lines, dtype=torch.int64, device=dev)
    d_nlen = torch.zeros(max_lines, dtype=torch.int32, device=dev)
    ws_bytes = lib.skh_parse_workspace_size(n, max_lines)
    ws_t, ws = aligned_ws(torch, ws_bytes, dev)
    d_res = torch.zeros(6, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    lib.check(lib.skh_parse_dev(ctx, d_text.data_ptr(), n, capi.FASTQ if fastq else capi.FASTA, 1, d_seq.data_ptr(),
                                d_qual.data_ptr(), padded, d_offs.data_ptr(), d_npos.data_ptr(), d_nlen.data_ptr(),
                                max_lines, ws, ws_bytes, max_lines, d_res.data_ptr(), None))
    lib.check(lib.skh_sync(ctx))
    res = capi.ParseResult.from_buffer_copy(d_res.cpu().numpy().tobytes())
    assert res.status == capi.PARSE_OK
    nrec, total = res.nrec, res.seq_bytes

    # the transform of seq.go:187-195
    if flags.get("reverse") or flags.get("complement"):
        # alphabet as the reader guesses it, from the first record (the checker's own guess function)
        first = np.ascontiguousarray(d_seq[:int(d_offs[1])].cpu().numpy()) if nrec else np.zeros(0, np.uint8)
        alphabet = env["ora"].dll.ora_guess_alphabet(first.ctypes.data_as(C.POINTER(C.c_uint8)), len(first), 10000)
        lut = lib.pair_table(alphabet) if flags.get("complement") else None
        so = torch.zeros_like(d_seq); qo = torch.zeros_like(d_qual)
        lib.check(lib.skh_revcomp_dev(ctx, d_seq.data_ptr(), so.data_ptr(), d_offs.data_ptr(), nrec, total, lut,
                                      1 if flags.get("reverse") else 0, None))
        if fastq and flags.get("reverse"):
            lib.check(lib.skh_revcomp_dev(ctx, d_qual.data_ptr(), qo.data_ptr(), d_offs.data_ptr(), nrec, total, None, 1, None))
            d_qual = qo
        d_seq = so

    # flag logic of seq.go:163-241
    first_len = int(d_offs[1] - d_offs[0]) if nrec else 0
    is_fastq = fastq and first_len > 0
    pn = ps = True
    pq = is_fastq
    plus = True
    if flags.get("only_name") and flags.get("only_seq"):
        pass
    elif flags.get("only_name"):
        ps = pq = False
    elif flags.get("only_seq"):
        pn = pq = False
    elif flags.get("only_qual"):
        pn = ps = False; pq = True; plus = False
    fmt = capi.SeqFormat(min_len=flags.get("min_len", -1), max_len=flags.get("max_len", -1),
                         line_width=0 if fastq else flags.get("width", 60), print_name=pn, print_seq=ps,
                         print_qual=pq, plus_line=plus, only_id=bool(flags.get("only_id")),
                         mark=b"@" if is_fastq else b">")
    cap = 2 * padded + 16 * (nrec + 1) + 64
    d_out = torch.full((cap,), 0xEE, dtype=torch.uint8, device=dev)
    fws_bytes = lib.skh_format_workspace_size(nrec)
    fws_t, fws = aligned_ws(torch, fws_bytes, dev)
    d_fres = torch.zeros(2, dtype=torch.int64, device=dev)
    lib.check(lib.skh_format_seq_dev(ctx, d_text.data_ptr(), d_npos.data_ptr(), d_nlen.data_ptr(), d_seq.data_ptr(),
                                     d_qual.data_ptr(), d_offs.data_ptr(), nrec, C.byref(fmt), d_out.data_ptr(), cap,
                                     fws, fws_bytes, d_fres.data_ptr(), None))
    lib.check(lib.skh_sync(ctx))
    fres = capi.FormatResult.from_buffer_copy(d_fres.cpu().numpy().tobytes())
    assert fres.status == capi.PARSE_OK
    assert bool((d_out[fres.out_bytes:] == 0xEE).all())
    return d_out[:fres.out_bytes].cpu().numpy().tobytes()


def cli_args(flags):
    a = []
    for k, f in (("reverse", "-r"), ("complement", "-p"), ("only_name", "-n"), ("only_seq", "-s"), ("only_qual", "-q"),
                 ("only_id", "-i")):
        if flags.get(k):
            a.append(f)
    if "width" in flags:
        a += ["-w", str(flags["width"])]
    if "min_len" in flags:
        a += ["-m", str(flags["min_len"])]
    if "max_len" in flags:
        a += ["-M", str(flags["max_len"])]
    return a


def check(env, tmp_path, text, flags, tag):
    path = str(tmp_path / ("in_%s.%s" % (tag, "fq" if text[:1] == b"@" else "fa")))
    with open(path, "wb") as fh:
        fh.write(text)
    want = subprocess.run([env["cli"], "seq"] + cli_args(flags) + [path], stdout=subprocess.PIPE, check=True).stdout
    got = gpu_seq_command(env, text, flags)
    assert len(got) == len(want)
    assert got == want


FLAG_SETS = [
    {}, {"width": 0}, {"width": 7}, {"width": 16}, {"width": 1},
    {"only_name": 1}, {"only_name": 1, "only_id": 1}, {"only_seq": 1}, {"only_seq": 1, "width": 0},
    {"only_name": 1, "only_seq": 1}, {"min_len": 60}, {"max_len": 80}, {"min_len": 50, "max_len": 100, "only_id": 1},
    {"reverse": 1, "complement": 1}, {"reverse": 1, "complement": 1, "width": 0}, {"reverse": 1}, {"complement": 1},
]


def load_fixture(name):
    p = os.path.join(GOLD, name)
    return gzip.open(p, "rb").read() if p.endswith(".gz") else open(p, "rb").read()


@pytest.mark.parametrize("k", range(len(FLAG_SETS)))
def test_hairpin_flags(env, tmp_path, k):
    check(env, tmp_path, load_fixture("hairpin.fa.gz"), FLAG_SETS[k], "hp%d" % k)


def test_hairpin_identity_roundtrip(env):
    """tests/test.sh:22-26 of the reference: `seq -w 60 hairpin.fa` reproduces the file byte for byte."""
    text = load_fixture("hairpin.fa.gz")
    assert gpu_seq_command(env, text, {"width": 60}) == text


@pytest.mark.parametrize("name", ["Illimina1.8.fq.gz", "reads_1.fq.gz", "Illimina1.5.fq"])
@pytest.mark.parametrize("flags", [{}, {"reverse": 1, "complement": 1}, {"only_qual": 1}, {"only_name": 1, "only_id": 1},
                                   {"only_seq": 1}, {"min_len": 100}, {"width": 10}])
def test_fastq_fixtures(env, tmp_path, name, flags):
    check(env, tmp_path, load_fixture(name), flags, name.split(".")[0])


@pytest.mark.parametrize("name", ["mature.fa.gz", "contigs.fa"])
def test_other_fasta_fixtures(env, tmp_path, name):
    for flags in ({}, {"reverse": 1, "complement": 1, "width": 70}, {"only_id": 1, "only_name": 1}):
        check(env, tmp_path, load_fixture(name), flags, name.split(".")[0])


EDGE = {
    "empty_file": b"",
    "empty_records": b">a\n>b x\n>c\nA\n>d\n",
    "exact_width": b">a\n" + b"ACGT" * 15 + b"\n>b\n" + b"ACGT" * 30 + b"\n>c\n" + b"A" * 59 + b"\n>d\n" + b"C" * 61 + b"\n",
    "space_names": b"> lead\nAC\n>\ttab first\nGT\n>x\ty z\nAA\n>\nCC\n",
    "long_record": b">L desc\n" + b"ACGTTGCA" * 20000 + b"\n>s\nAC\n",
    "fq_empty_seq": b"@r1\n\n+\n\n@r2\nA\n+\nI\n",
    "fq_first_empty": b"@r0\n\n+\n\n",
}


@pytest.mark.parametrize("case", sorted(EDGE))
@pytest.mark.parametrize("flags", [{}, {"width": 0}, {"width": 3}, {"only_id": 1, "only_name": 1}, {"reverse": 1, "complement": 1}])
def test_edge_cases(env, tmp_path, case, flags):
    check(env, tmp_path, EDGE[case], flags, case)


def test_random_fasta_many_widths(env, tmp_path):
    rng = np.random.default_rng(zlib.crc32(b"fmt"))
    parts = []
    for r in range(4000):
        L = int(rng.choice([0, 1, 15, 16, 17, 59, 60, 61, 119, 120, 121, int(rng.integers(0, 2000))]))
        parts.append(b">r%d d%d\n" % (r, rng.integers(0, 99)))
        if L:
            parts.append(np.frombuffer(b"ACGTN", np.uint8)[rng.integers(0, 5, L)].tobytes() + b"\n")
    text = b"".join(parts)
    for w in (0, 1, 15, 16, 17, 60, 61, 1000):
        check(env, tmp_path, text, {"width": w}, "rw%d" % w)


def test_capacity_error(env):
    torch, lib, capi, ctx = env["torch"], env["lib"], env["capi"], env["ctx"]
    dev = torch.device("cuda", 0)
    seq = torch.from_numpy(np.frombuffer(b"ACGTACGTACGTACGT", np.uint8).copy()).to(dev)
    offs = torch.tensor([0, 8, 16], dtype=torch.int64, device=dev)
    text = torch.from_numpy(np.frombuffer(b"name1 name2     ", np.uint8).copy()).to(dev)
    npos = torch.tensor([0, 6], dtype=torch.int64, device=dev)
    nlen = torch.tensor([5, 5], dtype=torch.int32, device=dev)
    fmt = capi.SeqFormat(min_len=-1, max_len=-1, line_width=0, print_name=1, print_seq=1, print_qual=0, plus_line=0,
                         only_id=0, mark=b">")
    out = torch.full((64,), 0xEE, dtype=torch.uint8, device=dev)
    wsb = lib.skh_format_workspace_size(2)
    ws_t, ws = aligned_ws(torch, wsb, dev)
    res_d = torch.zeros(2, dtype=torch.int64, device=dev)
    for cap, status in ((31, capi.PARSE_ECAPACITY), (32, capi.PARSE_OK)):
        lib.check(lib.skh_format_seq_dev(ctx, text.data_ptr(), npos.data_ptr(), nlen.data_ptr(), seq.data_ptr(), None,
                                         offs.data_ptr(), 2, C.byref(fmt), out.data_ptr(), cap, ws, wsb,
                                         res_d.data_ptr(), None))
        lib.check(lib.skh_sync(ctx))
        r = capi.FormatResult.from_buffer_copy(res_d.cpu().numpy().tobytes())
        assert r.status == status and r.out_bytes == 32
    assert out[:32].cpu().numpy().tobytes() == b">name1\nACGTACGT\n>name2\nACGTACGT\n"
