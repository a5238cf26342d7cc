"""CPU-side checks of the C-ABI library: it loads, exports every symbol the header declares,
and its host-only entry points (tables, merge rule, argument errors) behave."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    from seqkit_amd import build
    build.build_lib()
    from seqkit_amd import load
    return load()


def test_header_symbols_are_all_exported_and_bound(lib):
    from seqkit_amd.capi import SYMBOLS
    with open(os.path.join(ROOT, "include", "seqkit_hip.h")) as fh:
        text = re.sub(r"/\*.*?\*/", "", fh.read(), flags=re.S)
    declared = set(re.findall(r"\b(skh_[a-z0-9_]+)\s*\(", text))
    assert declared, "no declarations found"
    assert declared == set(SYMBOLS), (declared ^ set(SYMBOLS))
    for name in declared:
        assert hasattr(lib.dll, name), name


def test_version_and_padding(lib):
    assert b"gfx950" in lib.skh_version()
    assert [lib.skh_padded_size(n) for n in (0, 1, 16, 17)] == [0, 16, 16, 32]


def test_pair_tables_match_oracle(lib, oracle_lib):
    from helpers import Oracle
    ora = Oracle(oracle_lib)
    for a in range(6):
        assert bytes(lib.pair_table(a)) == ora.pair_table(a).tobytes()
    t = (C.c_uint8 * 256)()
    assert lib.skh_pair_table(9, t) < 0 and b"alphabet" in lib.skh_last_error()


def test_class_table(lib):
    t = bytes(lib.class_table(["G", "c", "AT", "-"]))
    assert t[ord("G")] == 0 and t[ord("g")] == 0 and t[ord("C")] == 1 and t[ord("c")] == 1
    assert t[ord("a")] == 2 and t[ord("T")] == 2 and t[ord("-")] == 3 and t[ord("N")] == 0xFF
    arr = (C.c_char_p * 2)(b"AC", b"c")
    out = (C.c_uint8 * 256)()
    assert lib.skh_class_table(arr, 2, out) < 0


def test_stats_merge_rule(lib):
    from seqkit_amd import StatsPartial
    a, b = StatsPartial(), StatsPartial()
    lib.skh_stats_init(C.byref(a)); lib.skh_stats_init(C.byref(b))
    assert a.as_tuple() == (0, 0, 0, 2**63 - 1, 0)
    b.num, b.len_sum, b.gap_sum, b.len_min, b.len_max = 3, 30, 2, 5, 15
    lib.skh_stats_merge(C.byref(a), C.byref(b))
    b.num, b.len_sum, b.gap_sum, b.len_min, b.len_max = 1, 7, 0, 7, 7
    lib.skh_stats_merge(C.byref(a), C.byref(b))
    assert a.as_tuple() == (4, 37, 2, 5, 15)


def test_init_without_gpu_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    ctx = C.c_void_p()
    assert lib.skh_init(0, C.byref(ctx)) == -2
    assert b"device" in lib.skh_last_error().lower()


def test_rccl_already_in_the_process_is_the_one_used(lib, tmp_path):
    """A process that already holds RCCL symbols (torch.distributed, a static link) must be given that copy,
    not a second one: a stand-in library with the nine entry points is loaded into the global scope first,
    and the library reports its version and that it did not open anything itself."""
    import subprocess
    import sys
    src = tmp_path / "fake_rccl.c"
    src.write_text("""
        int ncclGetVersion(int *v) { *v = 99999; return 0; }
        int ncclGetUniqueId(void *a) { return 1; }
        int ncclCommInitRank(void *a, int n, void *id, int r) { return 1; }
        int ncclCommInitAll(void *a, int n, const int *d) { return 1; }
        int ncclCommDestroy(void *a) { return 0; }
        int ncclAllReduce(const void *a, void *b, unsigned long n, int t, int o, void *c, void *s) { return 1; }
        int ncclGroupStart(void) { return 0; }
        int ncclGroupEnd(void) { return 0; }
        const char *ncclGetErrorString(int r) { return "stand-in"; }
    """)
    so = tmp_path / "libfake_rccl.so"
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-o", str(so), str(src)])
    code = ("import ctypes as C, sys\n"
            "sys.path.insert(0, %r)\n"
            "C.CDLL(%r, mode=C.RTLD_GLOBAL)\n"
            "from seqkit_amd import load\n"
            "lib = load()\n"
            "v = [C.c_int(-1), C.c_int(-1), C.c_int(-1)]\n"
            "rc = lib.skh_comm_rccl_version(C.byref(v[0]), C.byref(v[1]), C.byref(v[2]))\n"
            "print(rc, v[0].value, v[1].value > 0, v[2].value)\n" % (ROOT, str(so)))
    out = subprocess.check_output([sys.executable, "-c", code]).decode().split()
    assert out == ["0", "99999", "True", "1"], out
