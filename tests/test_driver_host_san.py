"""The HOST side of the driver (seqkit_amd/host/*.cpp: block pipeline, tail carry, multi-line FASTQ
rewrite, output ordering) under AddressSanitizer + UBSan and ThreadSanitizer, on the CPU.

tests/stub/ builds the driver's own sources against a CPU restatement of the C ABI (test
infrastructure that forwards the arithmetic to the oracle; never shipped).  What these tests prove is
about the host code only: no memory error, no undefined behaviour, no data race, and bytes equal to
the oracle CLI's.  GPU parity of the same command lines is tests/test_gpu_cli.py's business."""
import json
import os
import subprocess

import pytest

from conftest import GOLDEN, ROOT, check_vector, run_cli
from helpers import wrapped_fastq

STUB = os.path.join(ROOT, "tests", "stub")

with open(os.path.join(GOLDEN, "expected", "manifest.json")) as _fh:
    _NAMES = sorted(json.load(_fh))


@pytest.fixture(scope="session")
def san():
    subprocess.check_call(["make", "-C", STUB, "-s", "-j4"])
    os.environ["ASAN_OPTIONS"] = "detect_leaks=1:abort_on_error=0:exitcode=99"
    os.environ["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1:exitcode=98"
    os.environ["TSAN_OPTIONS"] = "halt_on_error=1:exitcode=97:report_signal_unsafe=0"
    return {"asan": os.path.join(STUB, "seqkit-hip-asan"), "tsan": os.path.join(STUB, "seqkit-hip-tsan")}


def clean(result):
    rc, _, err = result
    assert rc not in (97, 98, 99) and b"Sanitizer" not in err and b"runtime error" not in err, err.decode("latin-1")[-3000:]
    return result


@pytest.mark.parametrize("name", _NAMES)
def test_documented_vectors_under_asan(san, name, manifest):
    check_vector(san["asan"], name, manifest[name])


COMMANDS = [["seq", "-r", "-p"], ["seq", "-r", "-p", "-w", "0"], ["seq", "-n", "-i"], ["fx2tab", "-n", "-l", "-g", "-B", "AGCT"],
            ["fx2tab", "-a"], ["stats", "-a"], ["stats", "-T"], ["subseq", "-r", "2:-2"], ["seq", "-g", "-u"]]
FILES = ["hairpin.fa.gz", "reads_1.fq.gz", "contigs.fa", "Illimina1.5.fq"]


@pytest.mark.parametrize("which", ["asan", "tsan"])
@pytest.mark.parametrize("argv", COMMANDS, ids=lambda v: " ".join(v))
def test_commands_small_blocks(san, oracle_cli, argv, which):
    # blocks of a few records: every block boundary falls inside a record somewhere, tails are carried,
    # the three pipeline threads hand blocks round all the time
    for f in FILES:
        want = run_cli(oracle_cli, argv + [f])
        for extra in ([], ["--batch-bytes", "4096"], ["--batch-bytes", "30000"]):
            got = clean(run_cli(san[which], argv + extra + [f]))
            assert got[0] == want[0] and got[1] == want[1], (f, extra, got[2][-500:])


@pytest.mark.parametrize("which", ["asan", "tsan"])
def test_long_records_and_several_files(san, oracle_cli, which, tmp_path):
    # records much longer than a block (the block grows, the accumulator path), several files in a row
    import numpy as np
    rng = np.random.default_rng(3)
    big = tmp_path / "big.fa"
    with open(big, "wb") as fh:
        for i, n in enumerate([700000, 12, 0, 250000, 90000, 5]):
            s = rng.choice(np.frombuffer(b"ACGTacgtNn-", np.uint8), size=n).tobytes()
            fh.write(b">c%d some text\n" % i)
            for a in range(0, n, 70):
                fh.write(s[a:a + 70] + b"\n")
    for argv in (["seq", "-r", "-p"], ["fx2tab", "-n", "-l", "-g"], ["stats", "-a", "-T"]):
        files = [str(big), "hairpin.fa.gz", str(big)]
        want = run_cli(oracle_cli, argv + files)
        got = clean(run_cli(san[which], argv + ["--batch-bytes", "65536"] + files))
        assert got[0] == want[0] == 0 and got[1] == want[1], argv


@pytest.mark.parametrize("rewrite", ["auto", "host"])
@pytest.mark.parametrize("which", ["asan", "tsan"])
def test_multiline_fastq(san, oracle_cli, which, tmp_path, rewrite, monkeypatch):
    # auto: the library's parser of wrapped FASTQ (here the stub's) with the serial rewrite behind it; host: the
    # serial rewrite alone
    if rewrite != "auto":
        monkeypatch.setenv("SKH_FASTQ_REWRITE", rewrite)
    variants = {"wrapped": dict(), "late": dict(four_line_head=1500), "crlf": dict(crlf=True, width=37)}
    for name, kw in variants.items():
        f = str(tmp_path / (name + ".fq"))
        wrapped_fastq(f, **kw)
        for argv in (["seq", "-r", "-p", "-w", "0"], ["seq"], ["fx2tab", "-n", "-l", "-g", "-B", "AGCT"], ["stats", "-a"],
                     ["convert", "--to", "Illumina-1.5+"], ["subseq", "-r", "3:-3"]):
            want = run_cli(oracle_cli, argv + [f])
            for extra in ([], ["--batch-bytes", "20000"]):
                got = clean(run_cli(san[which], argv + extra + [f]))
                assert got[0] == want[0] == 0 and got[1] == want[1], (name, argv, extra, got[2][-500:])
    with open(f, "rb") as fh:
        text = fh.read().decode("latin-1")
    want = run_cli(oracle_cli, ["seq", "-r", "-p"], stdin=text)
    got = clean(run_cli(san[which], ["seq", "-r", "-p"], stdin=text))
    assert got[0] == 0 and got[1] == want[1]


@pytest.mark.parametrize("rewrite", ["auto", "host", "off"])
def test_multiline_fastq_small_cases(san, oracle_cli, rewrite, monkeypatch):
    if rewrite != "auto":
        monkeypatch.setenv("SKH_FASTQ_REWRITE", rewrite)
    cases = ["@r1\nACGT\nAC\n+\nIIII\nII\n", "@r1\nACGT\nAC\n+r1\n@III\n+I\n@r2\nAC\n+\nII", "@e\n+\n@r2\nAC\nGT\n+\nIIII\n",
             "\n\n@r1\nAC\n\nGT\n+\nII\nII\n\n", "@a\nAC\n+\nII\n\n\n@b\nACG\n+\nIII\n"]
    for text in cases:
        for argv in (["seq", "-r", "-p"], ["fx2tab", "-l"], ["stats", "-T"]):
            want = run_cli(oracle_cli, argv, stdin=text)
            got = clean(run_cli(san["asan"], argv, stdin=text))
            assert got[0] == want[0] == 0 and got[1] == want[1], (text, argv, got[2])
    for text in ("@r1\nACGT\n+\nIII\nII\n", "@r1\nACGT\nAC\n+\nIIII\n", "@r1\nACGT\n+\nIIII\nxx\n"):
        assert run_cli(oracle_cli, ["seq"], stdin=text)[0] == 255
        rc, _, err = clean(run_cli(san["asan"], ["seq"], stdin=text))
        assert rc == 255 and b"[ERRO]" in err


@pytest.mark.parametrize("which", ["asan", "tsan"])
def test_host_parse_mode(san, oracle_cli, which):
    for argv in (["seq", "-r", "-p"], ["fx2tab", "-n", "-l", "-g", "-B", "AT"], ["stats", "-a"]):
        for f in ("reads_2.fq.gz", "mature.fa.gz"):
            want = run_cli(oracle_cli, argv + [f])
            got = clean(run_cli(san[which], argv + ["--host-parse", "--batch-records", "100"] + [f]))
            assert got[0] == 0 and got[1] == want[1]


@pytest.mark.parametrize("which", ["asan", "tsan"])
def test_blocks_dealt_to_several_devices(san, oracle_cli, which, tmp_path, monkeypatch):
    # --gpus N: the blocks of a file go round N contexts; output order, tails and the merged stats
    # must not depend on N (the stub has as many "devices" as SKH_STUB_DEVICES says)
    monkeypatch.setenv("SKH_STUB_DEVICES", "8")
    f = str(tmp_path / "late.fq")
    wrapped_fastq(f, four_line_head=1800)
    for argv, files in ((["seq", "-r", "-p"], ["hairpin.fa.gz"]), (["seq"], ["reads_1.fq.gz", "contigs.fa"]),
                        (["fx2tab", "-n", "-l", "-g", "-B", "AGCT"], ["reads_2.fq.gz"]),
                        (["stats", "-a"], ["hairpin.fa.gz", "mature.fa.gz", "Illimina1.8.fq.gz", "contigs.fa"]),
                        (["stats", "-T"], [f]), (["seq", "-r", "-p", "-w", "0"], [f]),
                        (["convert", "--to", "Illumina-1.5+"], ["reads_1.fq.gz"]), (["subseq", "-r", "2:-2"], ["mature.fa.gz"])):
        want = run_cli(oracle_cli, argv + files)
        for dev in (["--gpus", "2"], ["--gpus", "5"], ["--devices", "3,3,1"]):
            got = clean(run_cli(san[which], argv + dev + ["--batch-bytes", "16384"] + files))
            assert got[0] == want[0] == 0 and got[1] == want[1], (argv, dev, got[2][-500:])
    rc, _, err = run_cli(san["asan"], ["stats", "--gpus", "9", "contigs.fa"])
    assert rc == 255 and b"[ERRO] device ordinal out of range" in err


def long_record_files(tmp_path):
    """one.fa: a single record of 1.2 Mb; mixed.fa: a long first record, short and empty ones, another long one."""
    import numpy as np
    rng = np.random.default_rng(4)
    paths = []
    for name, lens in (("one.fa", [1_200_003]), ("mixed.fa", [700_000, 12, 0, 250_000, 90_000, 5])):
        path = tmp_path / name
        with open(path, "wb") as fh:
            for i, n in enumerate(lens):
                s = rng.choice(np.frombuffer(b"ACGTacgtNn-", np.uint8), size=n).tobytes()
                fh.write(b">c%d some text\n" % i)
                for a in range(0, n, 60):
                    fh.write(s[a:a + 60] + b"\n")
        paths.append(str(path))
    return paths


@pytest.mark.parametrize("which", ["asan", "tsan"])
def test_one_batch_over_several_devices(san, oracle_cli, which, tmp_path, monkeypatch):
    # several GPUs: pinned batches (--host-parse, and any file whose first record is longer than a text
    # block) are cut over ALL devices, inside records where they are long; SKH_SHARES=1 shows the cuts
    monkeypatch.setenv("SKH_STUB_DEVICES", "8")
    monkeypatch.setenv("SKH_SHARES", "1")
    one, mixed = long_record_files(tmp_path)
    for argv in (["seq", "-r", "-p"], ["seq", "-p", "-w", "0"], ["fx2tab", "-n", "-l", "-g", "-B", "AGCT"], ["stats", "-a", "-T"]):
        for f in (one, mixed, "reads_1.fq.gz"):
            want = run_cli(oracle_cli, argv + [f])
            for dev in (["--gpus", "2"], ["--devices", "0,1,2"], ["--gpus", "8"]):
                for mode in ([], ["--host-parse", "--batch-records", "700"]):
                    got = clean(run_cli(san[which], argv + dev + mode + ["--batch-bytes", "65536"] + [f]))
                    assert got[0] == want[0] == 0 and got[1] == want[1], (argv, f, dev, mode, got[2][-500:])
                    shares = [l for l in got[2].decode().splitlines() if l.startswith("[shares]")]
                    routed = bool(mode) or f in (one, mixed)
                    assert bool(shares) == routed, (argv, f, mode)
                    if f == one and shares:
                        # every device worked on a piece of the one record
                        n = 2 if dev[1] == "2" else 3 if dev[1] == "0,1,2" else 8
                        assert len(shares) >= n and all("records 0..1" in l for l in shares[:n])
                        assert sum("head_cut 0 " not in l for l in shares[:n]) == n - 1


def test_fasta_after_fastq_keeps_the_fastq_layout(san, oracle_cli):
    # seq.go:140-149: the layout is decided once; records of a later FASTA file get "+" and an empty quality
    want = run_cli(oracle_cli, ["seq", "reads_1.fq.gz", "contigs.fa"])
    assert want[1].endswith(b"@length=9\nAAAAAAAAA\n+\n\n")
    for mode in ([], ["--host-parse"]):
        got = clean(run_cli(san["asan"], ["seq"] + mode + ["reads_1.fq.gz", "contigs.fa"]))
        assert got[0] == 0 and got[1] == want[1]


def test_stats_type_follows_seq_type_flag(san, oracle_cli):
    # stat.go:87,128-134: -t goes to the reader, the type column shows it
    for mode in ([], ["--host-parse"]):
        got = clean(run_cli(san["asan"], ["stats", "-T", "-t", "protein"] + mode + ["contigs.fa"]))
        want = run_cli(oracle_cli, ["stats", "-T", "-t", "protein", "contigs.fa"])
        assert got[1] == want[1] and b"\tProtein\t" in got[1]


def test_custom_id_regexp(san, oracle_cli, tmp_path):
    # root.go:68-69, helper.go:125-135: the ID is capture group 1 of --id-regexp, the whole head line
    # where it does not match; --id-ncbi is a fixed expression.  doc/docs/usage.md:345-348 pins one case
    # (golden vector G1f); the rest is compared with the oracle's restatement.
    ncbi = tmp_path / "ncbi.fa"
    ncbi.write_text(">gi|110645304|ref|NC_002516.2| Pseudomona\nACGTACGT\n>plain one\nAC\n>gi|9|ref|X_1.1| other\nGGCC\n")
    cases = [(["seq", "-n", "-i", "--id-regexp", r"^[^\s]+\s([^\s]+)\s"], "hairpin.fa.gz"),
             (["fx2tab", "-i", "-l", "--id-regexp", r"^[^\s]+\s([^\s]+)\s"], "mature.fa.gz"),
             (["seq", "-i", "--id-regexp", r"^(\w+)-"], "hairpin.fa.gz"),
             (["seq", "-n", "-i", "--id-ncbi"], str(ncbi)), (["fx2tab", "-n", "-i", "--id-regexp", r"^gi\|([^\|]+)\|"], str(ncbi)),
             (["seq", "-n", "-i", "--id-regexp", r"^([^\s]+)\s?"], "reads_1.fq.gz")]
    for argv, f in cases:
        want = run_cli(oracle_cli, argv + [f])
        for extra in ([], ["--batch-bytes", "8192"]):
            got = clean(run_cli(san["asan"], argv + extra + [f]))
            assert got[0] == want[0] == 0 and got[1] == want[1], (argv, extra)
    # \s of the default expression is Go's [\t\n\f\r ]: a vertical tab belongs to the ID
    vt = tmp_path / "vt.fa"
    vt.write_bytes(b">ab\x0bcd ef\nACGT\n>x\x0cy z\nAC\n")
    for argv in (["seq", "-n", "-i"], ["fx2tab", "-n", "-i"]):
        got = clean(run_cli(san["asan"], argv + [str(vt)]))
        assert got[1] == run_cli(oracle_cli, argv + [str(vt)])[1] and got[1].startswith(b"ab\x0bcd") and b"\nx" in got[1] and b"x\x0cy" not in got[1]
    rc, _, err = run_cli(san["asan"], ["seq", "-n", "-i", "--id-regexp", "abc", "contigs.fa"])
    assert rc == 255 and b"regular expression must contain" in err


def test_malformed_record_in_a_four_line_file_is_an_error_not_a_mode_switch(san, oracle_cli, tmp_path):
    # a quality line one byte short, far into a plain four-line file: the parse error names that record's
    # lines (the serial multi-line scan is for wrapped records only), whatever the block size
    import gzip
    with gzip.open(os.path.join(GOLDEN, "reads_1.fq.gz"), "rb") as fh:
        lines = fh.read().split(b"\n")
    bad = 4 * 1800 + 3
    lines[bad] = lines[bad][:-1]
    f = tmp_path / "short_quality.fq"
    f.write_bytes(b"\n".join(lines))
    assert run_cli(oracle_cli, ["seq", str(f)])[0] == 255
    for extra in ([], ["--batch-bytes", "20000"], ["--batch-bytes", "20000", "--gpus", "2"]):
        os.environ["SKH_STUB_DEVICES"] = "2"
        rc, out, err = clean(run_cli(san["asan"], ["seq", "-r", "-p"] + extra + [str(f)]))
        assert rc == 255 and b"[ERRO]" in err and b"invalid FASTA/Q format near line" in err, err
        n = int(err.decode().rsplit("near line", 1)[1].split()[0])
        assert bad - 3 <= n - 1 <= bad + 2, (n, bad)
    os.environ.pop("SKH_STUB_DEVICES", None)
