"""Shared fixtures.  GPU tests are marked `gpu`; everything else runs on CPU only."""
import gzip
import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def build_oracle():
    """Builds oracle/ (plain C, gcc) when its outputs are missing or stale."""
    odir = os.path.join(ROOT, "oracle")
    cli, lib = os.path.join(odir, "seqkit_oracle"), os.path.join(odir, "libseqkit_oracle.so")
    srcs = [os.path.join(odir, f) for f in ("seqkit_oracle.c", "seqkit_oracle.h", "oracle_cli.c")]
    newest = max(os.path.getmtime(s) for s in srcs)
    if not (os.path.exists(cli) and os.path.exists(lib)) or \
            min(os.path.getmtime(cli), os.path.getmtime(lib)) < newest:
        if shutil.which("gcc") or shutil.which("cc"):
            subprocess.check_call(["make", "-C", odir, "-s"])
    return cli, lib


@pytest.fixture(scope="session")
def oracle_cli():
    cli, _ = build_oracle()
    assert os.path.exists(cli), "oracle CLI missing and no C compiler to build it"
    return cli


@pytest.fixture(scope="session")
def oracle_lib():
    import ctypes
    _, lib = build_oracle()
    return ctypes.CDLL(lib)


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLDEN, "expected", "manifest.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def plain_fixtures(tmp_path_factory):
    """Decompressed copies of the .gz fixtures (path by base name without .gz)."""
    d = tmp_path_factory.mktemp("plain")
    out = {}
    for f in os.listdir(GOLDEN):
        src = os.path.join(GOLDEN, f)
        if f.endswith(".gz"):
            dst = os.path.join(d, f[:-3])
            with gzip.open(src, "rb") as a, open(dst, "wb") as b:
                shutil.copyfileobj(a, b)
            out[f[:-3]] = dst
        elif f.endswith((".fa", ".fq")):
            out[f] = src
    return out


def run_cli(cli, argv, stdin=None, cwd=GOLDEN):
    p = subprocess.run([cli] + list(argv), input=stdin.encode() if stdin is not None else None,
                       cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    return p.returncode, p.stdout, p.stderr


def check_vector(cli, name, vec):
    """Compares one documented known answer with what `cli` prints."""
    with open(os.path.join(GOLDEN, "expected", name + ".txt"), "rb") as fh:
        want = fh.read()
    rc, got, err = run_cli(cli, vec["argv"], vec.get("stdin"))
    if vec.get("pipe") and rc == 0:                      # `cmd ... | cmd <pipe argv>`
        rc, got, err = run_cli(cli, vec["pipe"], got.decode("latin-1"))
    mode = vec["mode"]
    if mode == "stderr255":
        assert rc == 255 and got == b"" and err == want
        return
    assert rc == 0, err.decode()
    if mode == "exact":
        assert got == want
    elif mode == "log+head4":
        lines = want.splitlines(True)
        log = b"".join(l for l in lines if l.startswith(b"["))
        assert err == log and b"".join(got.splitlines(True)[:4]) == b"".join(l for l in lines if not l.startswith(b"["))
    elif mode == "head3":
        assert b"".join(got.splitlines(True)[:3]) == want
    elif mode == "fields4":
        assert [l.split() for l in got.splitlines()[:4]] == [l.split() for l in want.splitlines()]
    elif mode == "expandtabs":
        assert got.decode().expandtabs(8).encode() == want
    elif mode == "head2_expandtabs":
        rows = [l.expandtabs(8).rstrip() for l in got.decode().splitlines()[:2]]
        assert rows == want.decode().splitlines()
    elif mode == "lengths":
        lens = [int(l.split(b"\t")[3]) for l in got.splitlines()]
        assert lens == [int(x) for x in want.split()]
    else:
        raise AssertionError("unknown mode " + mode)
