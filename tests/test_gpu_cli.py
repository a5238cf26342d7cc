"""End-to-end parity of the `seqkit-hip` driver: its bytes on stdout must equal the reference's
documented outputs and the CPU oracle's output for the same command line, on every fixture of
the reference's tests.  Both ways through the driver are covered: the default (file text parsed
and formatted on the GPU) and --host-parse (host reader -> CSR batches -> kernels -> host
formatter)."""
import json
import os

import pytest

from conftest import GOLDEN, ROOT, check_vector, run_cli
from helpers import wrapped_fastq

pytestmark = pytest.mark.gpu

CLI = os.path.join(ROOT, "seqkit_amd", "seqkit-hip")
GOLDEN_DIR = GOLDEN

with open(os.path.join(GOLDEN, "expected", "manifest.json")) as _fh:
    _NAMES = sorted(json.load(_fh))

FILES = ["hairpin.fa.gz", "mature.fa.gz", "Illimina1.8.fq.gz", "reads_1.fq.gz", "reads_2.fq.gz", "contigs.fa",
         "Illimina1.5.fq"]


@pytest.fixture(scope="module")
def cli():
    assert os.path.exists(CLI), "seqkit-hip is not built (python seqkit_amd/build.py)"
    return CLI


@pytest.mark.parametrize("name", _NAMES)
def test_documented_vector(cli, manifest, name):
    check_vector(cli, name, manifest[name])


SEQ_VARIANTS = [["-r", "-p"], ["-r"], ["-p"], [], ["-r", "-p", "-w", "0"], ["-rp", "-s"], ["-r", "-p", "-n", "-i"],
                ["-r", "-p", "-m", "100", "-M", "160"], ["-r", "-p", "-w", "7"]]


MODES = [[], ["--host-parse"]]


@pytest.mark.parametrize("mode", MODES, ids=["gpu-text", "host-parse"])
@pytest.mark.parametrize("f", FILES)
@pytest.mark.parametrize("flags", SEQ_VARIANTS, ids=lambda v: " ".join(v) or "plain")
def test_seq_matches_oracle(cli, oracle_cli, f, flags, mode):
    want = run_cli(oracle_cli, ["seq"] + flags + [f])
    got = run_cli(cli, ["seq"] + mode + flags + [f])
    assert got[0] == want[0] == 0 and got[1] == want[1]


@pytest.mark.parametrize("flags", [["-n"], ["-n", "-i"], ["-s", "-w", "0"], ["-n", "-s"], ["-q"], ["-m", "60", "-M", "90"]],
                         ids=lambda v: " ".join(v))
def test_seq_selection_flags_gpu_text(cli, oracle_cli, flags):
    for f in ("hairpin.fa.gz", "reads_1.fq.gz"):
        if flags == ["-q"] and f.endswith(".fa.gz"):
            continue
        want = run_cli(oracle_cli, ["seq"] + flags + [f])
        got = run_cli(cli, ["seq"] + flags + [f])
        assert got[0] == want[0] == 0 and got[1] == want[1], (f, flags)


@pytest.mark.parametrize("mode", MODES, ids=["gpu-text", "host-parse"])
def test_seq_validation(cli, oracle_cli, mode, tmp_path):
    # tests/test.sh:31-32 of the reference: RNA hairpins are not DNA
    rc, out, err = run_cli(cli, ["seq"] + mode + ["-t", "dna", "hairpin.fa.gz"])
    assert rc == 255 and out == b"" and b"invalid DNAredundant letter" in err
    # an invalid record in the middle: the records before it are printed, then the error
    p = tmp_path / "mid.fa"
    recs = [">r%d d\n%s\n" % (i, "ACGTN" * (5 + i % 7)) for i in range(500)]
    recs[321] = ">bad one\nACGTACGTeACGT\n"
    p.write_text("".join(recs))
    for extra in ([], ["--batch-bytes", "4096", "--batch-records", "64"], ["-r", "-p"]):
        flags = ["-v"] + [a for a in extra if a in ("-r", "-p")]
        want = run_cli(oracle_cli, ["seq"] + flags + [str(p)])
        got = run_cli(cli, ["seq"] + mode + ["-v"] + extra + [str(p)])
        assert want[0] == 255 and got[0] == 255 and got[1] == want[1]
        assert got[2].splitlines()[-1] == want[2].splitlines()[-1] == b"[ERRO] error when parsing seq: bad (invalid DNAredundant letter: e)"
    # beyond the validated prefix nothing is looked at; -V 0 looks at everything
    q = tmp_path / "far.fa"
    q.write_text(">long\n" + "A" * 12000 + "e" + "\n")
    assert run_cli(cli, ["seq"] + mode + ["-v", str(q)])[0] == 0
    assert run_cli(cli, ["seq"] + mode + ["-v", "-V", "0", str(q)])[0] == 255
    want = run_cli(oracle_cli, ["seq", "-v", "-V", "0", str(q)])
    assert want[0] == 255


MULTILINE_COMMANDS = [["seq"], ["seq", "-r", "-p", "-w", "0"], ["seq", "-r", "-p"], ["seq", "-n", "-i"], ["seq", "-q"],
                      ["fx2tab", "-n", "-l", "-g", "-B", "AGCT"], ["fx2tab"], ["stats"], ["stats", "-a", "-T"],
                      ["subseq", "-r", "3:-3"], ["convert", "--to", "Illumina-1.5+"]]


# who reads wrapped FASTQ: the GPU's parser of such text with the serial rewrite behind it for what it refuses (the
# default), the serial rewrite alone, the GPU's parser alone
REWRITE_MODES = ["auto", "host", "off"]


@pytest.mark.parametrize("rewrite", REWRITE_MODES)
@pytest.mark.parametrize("argv", MULTILINE_COMMANDS, ids=lambda v: " ".join(v))
def test_multiline_fastq_default_path(cli, oracle_cli, argv, tmp_path, rewrite, monkeypatch):
    # README.md:88, doc/docs/usage.md:357-359 of the reference: multi-line FASTQ is read like any other;
    # here without any flag, from files and from standard input, whatever the block size
    if rewrite != "auto":
        monkeypatch.setenv("SKH_FASTQ_REWRITE", rewrite)
    variants = {"wrapped": dict(), "late": dict(four_line_head=1500), "crlf": dict(crlf=True, width=37)}
    for name, kw in variants.items():
        f = str(tmp_path / (name + ".fq"))
        wrapped_fastq(f, **kw)
        want = run_cli(oracle_cli, argv + [f])
        assert want[0] == 0
        for extra in ([], ["--batch-bytes", "20000"]):
            got = run_cli(cli, argv + extra + [f])
            assert got[0] == 0 and got[1] == want[1], (name, extra, got[2])
        with open(f, "rb") as fh:
            text = fh.read().decode("latin-1")
        got = run_cli(cli, argv, stdin=text)
        want_stdin = run_cli(oracle_cli, argv, stdin=text)
        assert got[0] == 0 and got[1] == want_stdin[1], name


@pytest.mark.parametrize("rewrite", REWRITE_MODES)
def test_multiline_fastq_small_cases(cli, oracle_cli, rewrite, monkeypatch):
    if rewrite != "auto":
        monkeypatch.setenv("SKH_FASTQ_REWRITE", rewrite)
    cases = ["@r1\nACGT\nAC\n+\nIIII\nII\n",                       # the documented shape
             "@r1\nACGT\nAC\n+r1\n@III\n+I\n@r2\nAC\n+\nII",         # quality lines starting with '@' / '+', no final newline
             "@e\n+\n@r2\nAC\nGT\n+\nIIII\n",                        # a record without any sequence line
             "\n\n@r1\nAC\n\nGT\n+\nII\nII\n\n"]                     # blank lines inside the sequence
    for text in cases:
        for argv in (["seq", "-r", "-p"], ["fx2tab", "-l"], ["stats", "-T"]):
            want = run_cli(oracle_cli, argv, stdin=text)
            got = run_cli(cli, argv, stdin=text)
            assert got[0] == want[0] == 0 and got[1] == want[1], (text, argv, got[2])
    # quality longer or shorter than the sequence stays an error (exit status 255)
    for text in ("@r1\nACGT\n+\nIII\nII\n", "@r1\nACGT\nAC\n+\nIIII\n"):
        assert run_cli(oracle_cli, ["seq"], stdin=text)[0] == 255
        rc, _, err = run_cli(cli, ["seq"], stdin=text)
        assert rc == 255 and b"[ERRO]" in err


@pytest.mark.parametrize("rewrite", REWRITE_MODES)
def test_multiline_fastq_damaged_texts(cli, oracle_cli, rewrite, monkeypatch):
    """Wrapped FASTQ with a line deleted, doubled, inserted or re-headed, or cut at a byte: the driver fails where the
    oracle's CLI fails (exit status 255) and prints the same records where it does not, whoever reads the text."""
    import numpy as np
    if rewrite != "auto":
        monkeypatch.setenv("SKH_FASTQ_REWRITE", rewrite)
    rng = np.random.default_rng(123)
    n_ok = n_bad = 0
    for case in range(150):
        lines = []
        for r in range(int(rng.integers(1, 5))):
            L = int(rng.choice([0, 1, 7, 60, 61, 150]))
            s = "".join("ACGTN"[i] for i in rng.integers(0, 5, L))
            q = "".join(chr(c) for c in rng.choice([33, 43, 64, 70, 73], L))
            ws, wq = int(rng.choice([1, 13, 60, 1000])), int(rng.choice([1, 17, 60, 1000]))
            lines.append("@read%d some text" % r)
            lines += [s[a:a + ws] for a in range(0, L, ws)]
            lines.append("+")
            lines += [q[a:a + wq] for a in range(0, L, wq)]
            if rng.random() < 0.3:
                lines.append("")
        kind, k = case % 6, int(rng.integers(0, len(lines)))
        if kind == 0:
            del lines[k]
        elif kind == 1:
            lines.insert(k, lines[k])
        elif kind == 2:
            lines.insert(k, ["@x y", "+", "ACGT", "", "+@+@"][int(rng.integers(0, 5))])
        elif kind == 3 and lines[k]:
            lines[k] = "@+A"[int(rng.integers(0, 3))] + lines[k][1:]
        text = "\n".join(lines) + "\n"
        if kind == 4:
            text = text[:int(rng.integers(0, len(text) + 1))]
        want = run_cli(oracle_cli, ["seq", "-r", "-p"], stdin=text)
        for mode in MODES if rewrite == "auto" else MODES[:1]:
            got = run_cli(cli, ["seq", "-r", "-p"] + mode, stdin=text)
            assert want[0] in (0, 255) and got[0] == want[0], (case, kind, mode, text, got[2])
            if want[0] == 0:
                assert got[1] == want[1], (case, kind, mode, text)
            else:
                assert b"[ERRO]" in got[2]
        n_ok += want[0] == 0
        n_bad += want[0] != 0
    assert n_ok >= 30 and n_bad >= 30, (n_ok, n_bad)


def test_fasta_damaged_texts(cli, oracle_cli):
    """The same for FASTA: lines deleted, doubled, inserted, re-headed ('>', '@', ';', a base), text cut at a byte."""
    import numpy as np
    rng = np.random.default_rng(321)
    n_ok = n_bad = 0
    for case in range(150):
        lines = []
        for r in range(int(rng.integers(1, 5))):
            L = int(rng.choice([0, 1, 7, 60, 61, 150]))
            s = "".join("ACGTN>@"[i] for i in rng.choice(7, L, p=[.23, .23, .23, .23, .04, .02, .02]))
            w = int(rng.choice([1, 13, 60, 1000]))
            lines.append(">seq%d some text" % r)
            lines += [s[a:a + w] for a in range(0, L, w)]
            if rng.random() < 0.3:
                lines.append("")
        kind, k = case % 6, int(rng.integers(0, len(lines)))
        if kind == 0:
            del lines[k]
        elif kind == 1:
            lines.insert(k, lines[k])
        elif kind == 2:
            lines.insert(k, [">x y", "@x", "ACGT", "", ">"][int(rng.integers(0, 5))])
        elif kind == 3 and lines[k]:
            lines[k] = ">@;A"[int(rng.integers(0, 4))] + lines[k][1:]
        text = "\n".join(lines) + "\n"
        if kind == 4:
            text = text[:int(rng.integers(0, len(text) + 1))]
        for argv in (["seq", "-r", "-p", "-v", "-t", "dna"], ["fx2tab", "-n", "-l", "-g"]):
            want = run_cli(oracle_cli, argv, stdin=text)
            for mode in MODES:
                got = run_cli(cli, argv + mode, stdin=text)
                assert want[0] in (0, 255) and got[0] == want[0], (case, kind, mode, text, got[2])
                if want[0] == 0:
                    assert got[1] == want[1], (case, kind, mode, text)
        n_ok += want[0] == 0
        n_bad += want[0] != 0
    assert n_ok >= 30, (n_ok, n_bad)


@pytest.mark.parametrize("mode", MODES, ids=["gpu-text", "host-parse"])
def test_seq_small_batches_and_qual_only(cli, oracle_cli, mode):
    # batches / text blocks of a few records: results must not depend on where the input is cut
    for f in ("reads_1.fq.gz", "hairpin.fa.gz"):
        want = run_cli(oracle_cli, ["seq", "-r", "-p", f])
        got = run_cli(cli, ["seq"] + mode + ["-r", "-p", "--batch-records", "7", "--batch-bytes", "1000", f])
        assert got[1] == want[1]
    want = run_cli(oracle_cli, ["seq", "-r", "-q", "reads_2.fq.gz"])
    got = run_cli(cli, ["seq"] + mode + ["-r", "-q", "reads_2.fq.gz"])
    assert got[0] == 0 and got[1] == want[1]
    rc, _, err = run_cli(cli, ["seq"] + mode + ["-q", "contigs.fa"])
    assert rc == 255 and b"[ERRO] FASTA format has no quality" in err


FX_VARIANTS = [["-n", "-l", "-g", "-B", "AGCT"], ["-n", "-i", "-l", "-g", "-G", "-H"], ["-l", "-B", "a", "-B", "c", "-B", "ac"],
               ["-n", "-B", "AACC", "-B", "N", "-B", "-"], ["-n", "-g", "-B", "ACGTURYKMSWBDHVN-."], [],
               ["-n", "-a", "-l"]]


@pytest.mark.parametrize("mode", MODES, ids=["gpu-text", "host-parse"])
@pytest.mark.parametrize("f", FILES)
@pytest.mark.parametrize("flags", FX_VARIANTS, ids=lambda v: " ".join(v) or "plain")
def test_fx2tab_matches_oracle(cli, oracle_cli, f, flags, mode):
    want = run_cli(oracle_cli, ["fx2tab"] + flags + [f])
    got = run_cli(cli, ["fx2tab"] + mode + flags + [f])
    assert got[0] == want[0] == 0 and got[1] == want[1]


def test_fx2tab_small_blocks(cli, oracle_cli):
    for f in ("hairpin.fa.gz", "reads_1.fq.gz"):
        flags = ["-n", "-i", "-l", "-g", "-G", "-B", "AT", "-B", "N"]
        want = run_cli(oracle_cli, ["fx2tab"] + flags + [f])
        got = run_cli(cli, ["fx2tab", "--batch-bytes", "5000"] + flags + [f])
        assert got[0] == 0 and got[1] == want[1]


@pytest.mark.parametrize("flags", [["-a"], ["-n", "-i", "-a"], ["-n", "-l", "-g", "-B", "AT", "-a", "-H"]], ids=lambda v: " ".join(v))
def test_fx2tab_alphabet_column(cli, oracle_cli, flags, tmp_path):
    # fx2tab.go:145-147,167-180: the byte values present in each record, ascending (GPU rows)
    import numpy as np
    rng = np.random.default_rng(41)
    odd = tmp_path / "odd.fa"
    with open(odd, "wb") as fh:
        for i, n in enumerate([50, 0, 1, 40000, 300, 33000]):
            s = rng.choice(np.frombuffer(bytes(c for c in range(33, 127) if c != ord(">")), np.uint8), size=n).tobytes()
            fh.write(b">o%d\n" % i + s + b"\n")
    for f in FILES + [str(odd)]:
        for extra in ([], ["--batch-bytes", "50000"]):
            want = run_cli(oracle_cli, ["fx2tab"] + flags + [f])
            got = run_cli(cli, ["fx2tab"] + flags + extra + [f])
            assert got[0] == want[0] == 0 and got[1] == want[1], (f, flags, extra)


def test_fx2tab_float_trap(cli):
    # 34/160 + 33/160 prints 41.87 (two divisions, seqkit/cmd/fx2tab.go:128-133), not 41.88
    _, out, _ = run_cli(cli, ["fx2tab", "-n", "-i", "-l", "-g", "hairpin.fa.gz"])
    rows = {l.split(b"\t")[0]: l.split(b"\t") for l in out.splitlines()}
    assert rows[b"tcc-MIR172b"][3:] == [b"160", b"41.87"] and rows[b"cme-MIR167a"][3:] == [b"160", b"41.87"]


@pytest.mark.parametrize("mode", MODES, ids=["gpu-text", "host-parse"])
@pytest.mark.parametrize("flags", [[], ["-T"], ["-a"], ["-a", "-T"], ["-a", "-G", "N"]], ids=lambda v: " ".join(v) or "plain")
def test_stats_matches_oracle(cli, oracle_cli, flags, mode):
    want = run_cli(oracle_cli, ["stats"] + flags + FILES)
    got = run_cli(cli, ["stats"] + mode + flags + FILES)
    assert got[0] == want[0] == 0 and got[1] == want[1]
    got = run_cli(cli, ["stats"] + mode + ["--batch-bytes", "20000"] + flags + FILES)     # many blocks per file
    assert got[0] == 0 and got[1] == want[1]


def test_stdin_edge_cases(cli, oracle_cli):
    cases = [">a\nATAT\n>b\n>c\nGGC\n", ">s\nAC-GT..A C\n", ">x\nacgtnACGTN\n", "@r\nAACGN\n+\n!#%&(\n",
             ">only\n", ">a\r\nAC\r\nGT\r\n>b\r\nTTTT", ">e1\n>e2\n>e3\nA\n"]
    for text in cases:
        for argv in (["seq", "-r", "-p"], ["seq", "--host-parse", "-r", "-p"],
                     ["fx2tab", "-n", "-l", "-g", "-G", "-B", "AT"], ["fx2tab", "--host-parse", "-n", "-l", "-g", "-G", "-B", "AT"],
                     ["fx2tab", "-l", "-G"], ["stats", "-a", "-T"],
                     ["stats", "--host-parse", "-a", "-T"]):
            want = run_cli(oracle_cli, [a for a in argv if a != "--host-parse"], stdin=text)
            got = run_cli(cli, argv, stdin=text)
            assert got[0] == want[0] and got[1] == want[1], (text, argv)


def test_long_records_through_the_driver(cli, oracle_cli, tmp_path):
    import numpy as np
    rng = np.random.default_rng(5)
    p = tmp_path / "long.fa"
    with open(p, "wb") as fh:
        for i, n in enumerate([100000, 33, 1 << 20, 5, 250001]):
            s = rng.choice(np.frombuffer(b"ACGTNacgtn", np.uint8), size=n).tobytes()
            fh.write(b">r%d\n" % i)
            for a in range(0, n, 70):
                fh.write(s[a:a + 70] + b"\n")
    for argv in (["seq", "-r", "-p", str(p)], ["seq", "--host-parse", "-r", "-p", str(p)],
                 ["seq", "--batch-bytes", "65536", "-r", "-p", str(p)],          # blocks smaller than a record: they grow
                 ["fx2tab", "-n", "-l", "-g", "-B", "AT", str(p)], ["stats", "-a", str(p)],
                 ["stats", "--host-parse", "-a", str(p)]):
        want = run_cli(oracle_cli, [a for a in argv if a not in ("--host-parse", "--batch-bytes", "65536")])
        got = run_cli(cli, argv)
        assert got[0] == want[0] == 0 and got[1] == want[1]


def test_long_lines_and_long_reads(cli, oracle_cli, tmp_path):
    # records and text lines around and beyond the 32-KiB limit of the lane-group kernels: unwrapped
    # FASTA (one line per record), long FASTQ reads, every wrap width, fx2tab with sequence columns
    import numpy as np
    rng = np.random.default_rng(9)
    lens = [32767, 32768, 32769, 12, 65536 + 7, 0, 300001, 98304, 1, 20000, 30011, 8192, 16385]
    fa, fq = tmp_path / "unwrapped.fa", tmp_path / "long.fq"
    with open(fa, "wb") as fh, open(fq, "wb") as fh2:
        for i, n in enumerate(lens):
            s = rng.choice(np.frombuffer(b"ACGTNacgtnRY-", np.uint8), size=n).tobytes()
            q = rng.integers(33, 74, size=n, dtype=np.uint8).tobytes()
            fh.write(b">u%d some words\n" % i + s + b"\n")
            if n:
                fh2.write(b"@q%d x\n" % i + s + b"\n+\n" + q + b"\n")
    for f in (fa, fq):
        for argv in (["seq", "-r", "-p"], ["seq", "-w", "0"], ["seq", "-r", "-w", "7"], ["seq", "-p", "-w", "32768"],
                     ["seq", "-s", "-w", "100"], ["seq", "-r", "-p", "--batch-bytes", "100000"],
                     # widths around the limits of the writers' multiply-high line index (and far beyond any record)
                     ["seq", "-w", "8191"], ["seq", "-r", "-w", "8192"], ["seq", "-w", "8193"], ["seq", "-p", "-w", "10000"],
                     ["seq", "-w", "16384"], ["seq", "-w", "131072"], ["seq", "-w", "131073"], ["seq", "-w", "1000000"],
                     ["fx2tab"], ["fx2tab", "-l", "-g"], ["fx2tab", "-n", "-l", "-B", "AT"], ["stats", "-a", "-T"]):
            want = run_cli(oracle_cli, [a for a in argv if a not in ("--batch-bytes", "100000")] + [str(f)])
            got = run_cli(cli, argv + [str(f)])
            assert got[0] == want[0] == 0 and got[1] == want[1], (f.name, argv)
    want = run_cli(oracle_cli, ["seq", "-q", str(fq)])
    got = run_cli(cli, ["seq", "-q", str(fq)])
    assert got[0] == want[0] == 0 and got[1] == want[1]


def test_records_around_the_formatter_stage(cli, oracle_cli, tmp_path):
    # The seq / fx2tab writers and the parser's line copy put the text of consecutive records together in an LDS
    # stage (7 KiB of output, 3.5 KiB per plane) and fall back to direct writes for steps that do not fit:
    # record sizes on both sides of those limits, mixed with empty and tiny records so that steps of every
    # fill alternate with steps that do not fit, names longer than a chunk, FASTA in several widths and FASTQ.
    import numpy as np
    rng = np.random.default_rng(77)
    sizes = [0, 1, 3, 15, 16, 17, 40, 150, 151, 1700, 3400, 3500, 3583, 3584, 3585, 3700, 6900, 7000, 7100, 7168, 7200, 14500]
    fa, fq = tmp_path / "stage.fa", tmp_path / "stage.fq"
    with open(fa, "wb") as fh, open(fq, "wb") as fh2:
        for i in range(2500):
            n = int(rng.choice(sizes)) if rng.random() < 0.3 else int(rng.integers(0, 400))
            s = rng.choice(np.frombuffer(b"ACGTNacgtn-", np.uint8), size=n).tobytes()
            q = rng.integers(33, 74, size=n, dtype=np.uint8).tobytes()
            name = b"r%d %s" % (i, b"w" * int(rng.integers(0, 70)))
            fh.write(b">" + name + b"\n")
            for a in range(0, n, 80):
                fh.write(s[a:a + 80] + b"\n")
            if n:
                fh2.write(b"@" + name + b"\n" + s + b"\n+\n" + q + b"\n")
    for f in (fa, fq):
        for argv in (["seq"], ["seq", "-w", "0"], ["seq", "-r", "-p", "-w", "61"], ["seq", "-n"], ["seq", "-n", "-i"], ["seq", "-s", "-w", "16"],
                     ["seq", "-m", "3500"], ["seq", "-M", "3584", "-w", "7000"], ["seq", "--batch-bytes", "70000"],
                     ["fx2tab"], ["fx2tab", "-n", "-l", "-g", "-B", "AT"], ["fx2tab", "-l", "-G", "-a"], ["stats", "-a", "-T"]):
            want = run_cli(oracle_cli, [a for a in argv if a not in ("--batch-bytes", "70000")] + [str(f)])
            got = run_cli(cli, argv + [str(f)])
            assert got[0] == want[0] == 0 and got[1] == want[1], (f.name, argv)


SEQ_EDIT_FLAGS = [["-g"], ["-g", "-u"], ["-g", "-G", "N-"], ["-r", "-p", "-g"], ["-l"], ["-u", "-r"], ["--dna2rna"],
                  ["--rna2dna"], ["--rna2dna", "-p", "-u"], ["--dna2rna", "-s", "-w", "0"], ["-g", "-m", "100", "-M", "160"],
                  ["-g", "-G", "ACGTUacgtu", "-m", "60"], ["-g", "-q"], ["-l", "-n"], ["-g", "-r", "-p", "-l", "--dna2rna"]]


@pytest.mark.parametrize("flags", SEQ_EDIT_FLAGS, ids=lambda v: " ".join(v))
def test_seq_gaps_conversions_and_case(cli, oracle_cli, flags, tmp_path):
    # the transforms behind reverse / complement in seq's record loop (seq.go:196-241)
    import numpy as np
    rng = np.random.default_rng(31)
    gapped = tmp_path / "gapped.fa"
    with open(gapped, "wb") as fh:
        for i, n in enumerate([70, 0, 150, 33000, 5, 1200, 160, 99, 100, 101]):
            s = rng.choice(np.frombuffer(b"ACGTacgtNn-. -UuRY", np.uint8), size=n).tobytes()
            fh.write(b">g%d with gaps\n" % i)
            for a in range(0, max(n, 1), 60):
                fh.write(s[a:a + 60] + b"\n")
    gq = tmp_path / "gapped.fq"
    with open(gq, "wb") as fh:
        for i, n in enumerate([70, 150, 40000, 5, 99, 100]):
            s = rng.choice(np.frombuffer(b"ACGTN-.", np.uint8), size=n).tobytes()
            q = rng.integers(33, 74, size=n, dtype=np.uint8).tobytes()
            fh.write(b"@q%d\n" % i + s + b"\n+\n" + q + b"\n")
    for f in ("hairpin.fa.gz", "reads_1.fq.gz", str(gapped), str(gq)):
        if "-q" in flags and not f.endswith(("fq", "fq.gz")):
            continue
        for extra in ([], ["--batch-bytes", "30000"]):
            want = run_cli(oracle_cli, ["seq"] + flags + [f])
            got = run_cli(cli, ["seq"] + flags + extra + [f])
            assert got[0] == want[0] == 0 and got[1] == want[1] and got[2] == want[2], (f, flags, extra)


def test_seq_edit_flag_errors(cli, oracle_cli):
    for argv in (["seq", "-l", "-u", "contigs.fa"], ["seq", "-g", "-G", "", "contigs.fa"]):
        want = run_cli(oracle_cli, argv)
        got = run_cli(cli, argv)
        assert got[0] == want[0] == 255 and got[2] == want[2] and got[1] == b""


def test_large_output_to_regular_files(cli, oracle_cli, tmp_path):
    # tens of megabytes of output into regular files: -o, stdout redirected to a file that already
    # holds text, and a file opened for appending
    import subprocess
    import numpy as np
    rng = np.random.default_rng(21)
    n, L = 120_000, 150
    seq = np.frombuffer(b"ACGTN", np.uint8)[rng.integers(0, 5, (n, L))]
    qual = rng.integers(35, 74, (n, L), dtype=np.uint8)
    fq = tmp_path / "big.fq"
    with open(fq, "wb") as fh:
        for i in range(n):
            fh.write(b"@r%d extra\n" % i + seq[i].tobytes() + b"\n+\n" + qual[i].tobytes() + b"\n")
    want = run_cli(oracle_cli, ["seq", "-r", "-p", str(fq)])
    assert want[0] == 0 and len(want[1]) > (32 << 20)
    out1 = tmp_path / "o1.fq"
    assert run_cli(cli, ["seq", "-r", "-p", "-o", str(out1), str(fq)])[0] == 0
    assert out1.read_bytes() == want[1]
    out2 = tmp_path / "o2.fq"
    with open(out2, "wb") as fh:
        fh.write(b"before\n")
        fh.flush()
        assert subprocess.run([cli, "seq", "-r", "-p", str(fq)], stdout=fh, timeout=300).returncode == 0
    assert out2.read_bytes() == b"before\n" + want[1]
    out3 = tmp_path / "o3.fq"
    out3.write_bytes(b"before\n")
    with open(out3, "ab") as fh:
        assert subprocess.run([cli, "seq", "-r", "-p", str(fq)], stdout=fh, timeout=300).returncode == 0
    assert out3.read_bytes() == b"before\n" + want[1]
    tab = run_cli(oracle_cli, ["fx2tab", str(fq)])
    out4 = tmp_path / "o4.tsv"
    assert run_cli(cli, ["fx2tab", "-o", str(out4), str(fq)])[0] == 0
    assert out4.read_bytes() == tab[1]


# ---------------------------------------------------------------------------------------- subseq

@pytest.mark.parametrize("f", FILES)
@pytest.mark.parametrize("region", ["1:12", "-12:-1", "13:-13", "1:-1", "5:5", "200:100", "-3:-5", "30:1000000"])
def test_subseq_region_matches_oracle(cli, oracle_cli, f, region):
    want = run_cli(oracle_cli, ["subseq", "-r", region, f])
    got = run_cli(cli, ["subseq", "-r", region, f])
    assert got[0] == want[0] == 0 and got[1] == want[1]
    got = run_cli(cli, ["subseq", "--batch-bytes", "4096", "-r", region, "-w", "7", f])      # many blocks per file
    want = run_cli(oracle_cli, ["subseq", "-r", region, "-w", "7", f])
    assert got[0] == 0 and got[1] == want[1]


def test_subseq_region_errors(cli, oracle_cli):
    for argv in (["subseq", "-r", "0:5"], ["subseq", "-r", "-3:4"], ["subseq", "-r", "abc"], ["subseq", "-r", "1:5", "-u", "3"],
                 ["subseq", "--gtf", "subseq_test.gtf", "-f"], ["subseq", "--gtf", "subseq_test.gtf", "-f", "-u", "1", "-d", "1"]):
        want = run_cli(oracle_cli, argv + ["contigs.fa"])
        got = run_cli(cli, argv + ["contigs.fa"])
        assert got[0] == want[0] == 255 and got[1] == want[1] == b"" and got[2] == want[2], argv


def _feature_files(tmp_path, ids, rng, lengths):
    gtf, bed = [], []
    types = ["CDS", "exon", "gene"]
    for k in range(400):
        i = int(rng.integers(0, len(ids)))
        a = int(rng.integers(1, max(lengths[i], 1) + 5))
        b = a + int(rng.integers(1, 60))
        strand = "+-."[int(rng.integers(0, 3))]
        gtf.append("%s\tsrc\t%s\t%d\t%d\t.\t%s\t.\tgene_id \"g%d\"; transcript_id \"t%d\";\n"
                   % (ids[i], types[k % 3], a, b, strand, k, k))
        cols = [ids[i], str(a - 1), str(b)]
        if k % 4:
            cols.append("name%d" % k)
            if k % 2:
                cols += ["0", "+-"[k % 3 == 0]]
        bed.append("\t".join(cols) + "\n")
    g, b = tmp_path / "f.gtf", tmp_path / "f.bed"
    g.write_text("# comment\n" + "".join(gtf)); b.write_text("track name=x\n" + "".join(bed))
    return str(g), str(b)


@pytest.mark.parametrize("f", ["hairpin.fa.gz", "reads_1.fq.gz", "contigs.fa"])
def test_subseq_features_match_oracle(cli, oracle_cli, f, tmp_path):
    import numpy as np
    from helpers import read_fastx_plain
    names, _, _, offs = read_fastx_plain(os.path.join(GOLDEN_DIR, f))
    ids = [n.split()[0].decode() for n in names][:300]
    lens = [int(offs[i + 1] - offs[i]) for i in range(len(ids))]
    rng = np.random.default_rng(len(f))
    gtf, bed = _feature_files(tmp_path, ids, rng, lens)
    for flags in ([], ["-u", "3", "-d", "2"], ["-u", "100", "-d", "100"], ["-u", "5", "-f"], ["-d", "4", "-f"], ["-w", "11"]):
        for sel in (["--gtf", gtf], ["--bed", bed], ["--gtf", gtf, "--feature", "cds,GENE"], ["--gtf", gtf, "--chr", ids[0] + "," + ids[5]],
                    ["--gtf", gtf, "--gtf-tag", "transcript_id"]):
            want = run_cli(oracle_cli, ["subseq"] + sel + flags + [f])
            got = run_cli(cli, ["subseq"] + sel + flags + [f])
            assert got[0] == want[0] == 0 and got[1] == want[1], (sel, flags)
    got = run_cli(cli, ["subseq", "--batch-bytes", "4096", "--gtf", gtf, "-u", "7", f])
    want = run_cli(oracle_cli, ["subseq", "--gtf", gtf, "-u", "7", f])
    assert got[0] == 0 and got[1] == want[1]


def test_subseq_long_overlapping_features(cli, oracle_cli, tmp_path):
    import numpy as np
    rng = np.random.default_rng(9)
    fa = tmp_path / "chr.fa"
    with open(fa, "wb") as fh:
        for i, n in enumerate([300000, 1000, 120000]):
            s = rng.choice(np.frombuffer(b"ACGTNacgtn", np.uint8), size=n).tobytes()
            fh.write(b">chr%d some text\n" % i)
            for a in range(0, n, 60):
                fh.write(s[a:a + 60] + b"\n")
    bed = tmp_path / "big.bed"
    rows = []
    for k in range(60):                                 # far more selected bases than the file holds
        rows.append("chr%d\t%d\t%d\tf%d\t0\t%s\n" % (k % 3, k * 10, 100000 + k * 1000, k, "+-"[k & 1]))
    bed.write_text("".join(rows))
    want = run_cli(oracle_cli, ["subseq", "--bed", str(bed), str(fa)])
    got = run_cli(cli, ["subseq", "--bed", str(bed), str(fa)])
    assert got[0] == want[0] == 0 and got[1] == want[1]
    got = run_cli(cli, ["subseq", "--bed", str(bed), "--batch-bytes", "65536", str(fa)])
    assert got[0] == 0 and got[1] == want[1]


# ---------------------------------------------------------------------------------------- convert

ENCODINGS = ["Sanger", "Solexa", "Illumina-1.3+", "Illumina-1.5+", "Illumina-1.8+"]


@pytest.mark.parametrize("f", ["Illimina1.8.fq.gz", "Illimina1.5.fq", "reads_1.fq.gz"])
@pytest.mark.parametrize("to", ENCODINGS)
def test_convert_guessed_matches_oracle(cli, oracle_cli, f, to):
    for extra in ([], ["-f"], ["-n", "5"], ["-d"], ["--batch-bytes", "4096"]):
        want = run_cli(oracle_cli, ["convert", "--to", to] + [a for a in extra if not a.startswith("--batch") and a != "4096"] + [f])
        got = run_cli(cli, ["convert", "--to", to] + extra + [f])
        assert got == want, (f, to, extra)


@pytest.mark.parametrize("frm", ENCODINGS)
@pytest.mark.parametrize("to", ENCODINGS)
def test_convert_explicit_matches_oracle(cli, oracle_cli, frm, to):
    # every byte value a quality line can hold, so that invalid scores and every table entry are met
    import random
    rnd = random.Random(len(frm) * 7 + len(to))
    lo = 33 if frm in ("Sanger", "Illumina-1.8+") else (59 if frm == "Solexa" else 64)
    recs = []
    for i in range(300):
        n = rnd.randint(0, 120)
        q = "".join(chr(rnd.randint(lo, 126)) for _ in range(n))
        if i == 250:
            q = "!" * 5 + q                       # below the minimum of the 64-offset encodings
            n += 5
        recs.append("@r%d x\n%s\n+\n%s\n" % (i, "".join(rnd.choice("ACGTN") for _ in range(n)), q))
    text = "".join(recs)
    for extra in ([], ["-f"]):
        want = run_cli(oracle_cli, ["convert", "--from", frm, "--to", to] + extra, stdin=text)
        got = run_cli(cli, ["convert", "--from", frm, "--to", to] + extra, stdin=text)
        assert got == want, (frm, to, extra)


def test_convert_rejects_fasta(cli, oracle_cli):
    want = run_cli(oracle_cli, ["convert", "contigs.fa"])
    got = run_cli(cli, ["convert", "contigs.fa"])
    assert got[0] == want[0] == 255 and got[1] == want[1] == b"" and got[2] == want[2]
    assert run_cli(cli, ["convert", "--to", "phred", "reads_1.fq.gz"])[0] == 255


def test_convert_files_through_the_block_pipeline(cli, oracle_cli, tmp_path):
    # regular files go through the pipelined reader (the guess is made on a first, small read of
    # the file), standard input through the one-pass reader: same bytes, same messages, also with
    # an invalid score in the middle of a file, a file shorter than -n, and blocks much smaller
    # than the files
    import random
    rnd = random.Random(77)

    def fastq(nrec, lo, hi, bad_at=None):
        recs = []
        for i in range(nrec):
            n = rnd.randint(0, 160)
            q = "".join(chr(rnd.randint(lo, hi)) for _ in range(n))
            if i == bad_at:
                q, n = "!!!" + q, n + 3
            recs.append("@r%d x\n%s\n+\n%s\n" % (i, "".join(rnd.choice("ACGTN") for _ in range(n)), q))
        return "".join(recs)

    short = tmp_path / "short.fq"
    short.write_text(fastq(40, 33, 73))
    long_ = tmp_path / "long.fq"
    long_.write_text(fastq(5000, 33, 73))
    bad = tmp_path / "bad.fq"
    bad.write_text(fastq(3000, 64, 104, bad_at=2100))
    to13 = ["convert", "--to", "Illumina-1.3+"]
    for args in (["convert", str(long_)],
                 to13 + [str(long_)],
                 to13 + [str(short)],
                 to13 + [str(long_), str(short)],
                 to13 + ["-n", "4000", str(long_), str(short)],
                 ["convert", "--from", "Illumina-1.3+", str(bad), str(long_)],
                 to13 + ["-n", "50", str(long_), str(bad)],
                 to13 + ["-n", "40", str(short)],             # exactly -n records: the reference prints nothing
                 to13 + ["-n", "39", str(short)],
                 to13 + ["-n", "41", str(short)],
                 to13 + ["-d", str(short), str(long_)]):
        want = run_cli(oracle_cli, args)
        for extra in ([], ["--batch-bytes", "16384"]):
            got = run_cli(cli, args + extra)
            assert got == want, (args, extra)
    text = long_.read_text()
    want = run_cli(oracle_cli, ["convert", "-n", "4000"], stdin=text)
    for extra in ([], ["--batch-bytes", "16384"]):
        assert run_cli(cli, ["convert", "-n", "4000"] + extra, stdin=text) == want
    text = short.read_text()
    for n in ("39", "40", "41"):
        assert run_cli(cli, to13 + ["-n", n], stdin=text) == run_cli(oracle_cli, to13 + ["-n", n], stdin=text)


def test_custom_id_regexp(cli, oracle_cli, tmp_path):
    # --id-regexp / --id-ncbi (root.go:68-69): IDs cut out on the host, printed by the GPU formatter
    ncbi = tmp_path / "ncbi.fa"
    ncbi.write_text(">gi|110645304|ref|NC_002516.2| Pseudomona\nACGTACGT\n>plain one\nAC\n>gi|9|ref|X_1.1| other\nGGCC\n")
    cases = [(["seq", "-n", "-i", "--id-regexp", r"^[^\s]+\s([^\s]+)\s"], "hairpin.fa.gz"),
             (["fx2tab", "-i", "-l", "-g", "--id-regexp", r"^[^\s]+\s([^\s]+)\s"], "mature.fa.gz"),
             (["seq", "-i", "-r", "-p", "--id-regexp", r"^(\w+)-"], "hairpin.fa.gz"),
             (["seq", "-n", "-i", "--id-ncbi"], str(ncbi)), (["fx2tab", "-n", "-i", "--id-regexp", r"^gi\|([^\|]+)\|"], str(ncbi))]
    for argv, f in cases:
        want = run_cli(oracle_cli, argv + [f])
        for extra in ([], ["--batch-bytes", "8192"], ["--devices", "0,0", "--batch-bytes", "8192"]):
            got = run_cli(cli, argv + extra + [f])
            assert got[0] == want[0] == 0 and got[1] == want[1], (argv, extra)



def test_id_white_space_is_go_white_space(cli, oracle_cli, tmp_path):
    # \s of the default `^([^\s]+)\s?` is Go's [\t\n\f\r ]: a vertical tab belongs to the ID, a form feed ends it
    # (no reference-held vector has such a head line: the rule is RE2's documented class)
    vt = tmp_path / "vt.fa"
    vt.write_bytes(b">ab\x0bcd ef\nACGT\n>x\x0cy z\nAC\n")
    for argv in (["seq", "-n", "-i"], ["fx2tab", "-n", "-i"], ["seq", "-i", "-r"]):
        for mode in ([], ["--host-parse"]):
            got = run_cli(cli, argv + mode + [str(vt)])
            assert got[0] == 0 and got[1] == run_cli(oracle_cli, argv + [str(vt)])[1]
            assert b"ab\x0bcd" in got[1] and b"x\x0cy" not in got[1]
