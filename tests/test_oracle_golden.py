"""Pins the CPU oracle against every known answer the reference's docs and tests hold
for the hot path (reverse-complement, base content, stats)."""
import hashlib
import json
import os

import pytest

from conftest import GOLDEN, check_vector, run_cli

with open(os.path.join(GOLDEN, "expected", "manifest.json")) as _fh:
    _NAMES = sorted(json.load(_fh))


@pytest.mark.parametrize("name", _NAMES)
def test_documented_vector(oracle_cli, manifest, name):
    check_vector(oracle_cli, name, manifest[name])


def test_seq_w60_reproduces_input(oracle_cli, plain_fixtures):
    # tests/test.sh:22-26: `seq -w 60 hairpin.fa` is byte-identical to its input
    rc, got, _ = run_cli(oracle_cli, ["seq", "-w", "60", plain_fixtures["hairpin.fa"]])
    assert rc == 0
    with open(plain_fixtures["hairpin.fa"], "rb") as fh:
        want = fh.read()
    assert hashlib.md5(want).hexdigest() == "c08cd6d5ccaa6049fe161eaaea6e315a"
    assert got == want


def test_reverse_then_complement_equals_both(oracle_cli, plain_fixtures, tmp_path):
    # tests/test.sh:79-84: `seq -r | seq -p` == `seq -r -p`
    f = plain_fixtures["hairpin.fa"]
    _, r, _ = run_cli(oracle_cli, ["seq", "-r", f])
    p = tmp_path / "r.fa"
    p.write_bytes(r)
    _, rp2, _ = run_cli(oracle_cli, ["seq", "-p", str(p)])
    _, rp, _ = run_cli(oracle_cli, ["seq", "-r", "-p", f])
    assert rp == rp2 and len(rp) == 4720374


def _revcomp_of(oracle_cli, seq, tmp_path):
    p = tmp_path / "t.fa"
    p.write_text(">seq\n%s\n" % seq)
    _, out, _ = run_cli(oracle_cli, ["seq", "-r", "-p", "-s", "-w", "0", str(p)])
    return out.decode().strip()


def test_small_revcomp_known_answers(oracle_cli, tmp_path):
    # tests/test.sh:142-160 (subseq --gtf, minus strand = RevCom of the slice of acgtnACGTN)
    assert _revcomp_of(oracle_cli, "tnA", tmp_path) == "Tna"
    assert _revcomp_of(oracle_cli, "acgtnACG", tmp_path) == "CGTnacgt"
    assert _revcomp_of(oracle_cli, "acgtnACGTN", tmp_path) == "NACGTnacgt"
    # the -u 3 -d 2 case: minus-strand feature 4..6 with 3 up / 2 down = revcom(cgtnACGT)
    assert _revcomp_of(oracle_cli, "cgtnACGT", tmp_path) == "ACGTnacg"
    # doc/docs/usage.md:455-496
    assert _revcomp_of(oracle_cli, "ACTG", tmp_path) == "CAGT"
    assert _revcomp_of(oracle_cli, "ACTGact", tmp_path) == "agtCAGT"
    assert _revcomp_of(oracle_cli, "act", tmp_path) == "agt"


def test_fastq_reverse_also_reverses_quality(oracle_cli):
    # doc/docs/download.md:156 (v0.4.5): -r reverses the quality string with the bases
    rc, out, _ = run_cli(oracle_cli, ["seq", "-r", "-p"], stdin="@r\nAACGN\n+\n!#%&(\n")
    assert rc == 0 and out == b"@r\nNCGTT\n+\n(&%#!\n"


def test_gc_float_trap_records(oracle_cli):
    # g and c are divided by the length separately before being added
    # (seqkit/cmd/fx2tab.go:128-133): 34/160 + 33/160 prints 41.87, not 41.88.
    rc, out, _ = run_cli(oracle_cli, ["fx2tab", "-n", "-i", "-l", "-g", "hairpin.fa.gz"])
    rows = {l.split(b"\t")[0]: l.split(b"\t") for l in out.splitlines()}
    for rid in (b"tcc-MIR172b", b"cme-MIR167a"):
        assert rows[rid][3] == b"160" and rows[rid][4] == b"41.87"


def test_faq_column_layout(oracle_cli):
    # doc/docs/faq.md:69-73 layout: name, two empty columns, then one %.2f per -B value
    rc, out, _ = run_cli(oracle_cli, ["fx2tab", "-H", "-n", "-i", "-B", "a", "-B", "c", "-B", "ac"],
                         stdin=">x y\nAACCGT\n")
    assert out == b"#name\tseq\tqual\ta\tc\tac\nx\t\t\t33.33\t33.33\t66.67\n"


def test_gc_skew_nan_and_empty_record(oracle_cli):
    rc, out, _ = run_cli(oracle_cli, ["fx2tab", "-n", "-l", "-g", "-G"], stdin=">a\nATAT\n>b\n>c\nGGC\n")
    assert out == b"a\t\t\t4\t0.00\tNaN\nb\t\t\t0\t0.00\tNaN\nc\t\t\t3\t100.00\t33.33\n"


def test_contigs_and_gaps(oracle_cli):
    rc, out, _ = run_cli(oracle_cli, ["stats", "-a", "-T", "contigs.fa"])
    assert out.splitlines()[1] == b"contigs.fa\tFASTA\tDNA\t9\t54\t2\t6.0\t10\t3\t6\t8\t0\t8"
    rc, out, _ = run_cli(oracle_cli, ["stats", "-a", "-T"], stdin=">s\nAC-GT..A C\n")
    assert out.splitlines()[1].split(b"\t")[11] == b"4"
