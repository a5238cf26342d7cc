"""The sharding rules of the C ABI (skh_split_by_bytes, skh_split_record: host-only code of the
library, no GPU needed) against the Python statement of the same rules used by the gloo tests."""
import ctypes as C

import numpy as np
import pytest

from seqkit_amd import shard


@pytest.fixture(scope="module")
def lib():
    from seqkit_amd import build, load
    build.build_lib()
    return load()


def c_split_by_bytes(lib, offs, nparts):
    cuts = np.zeros(nparts + 1, np.uint64)
    lib.check(lib.skh_split_by_bytes(offs.ctypes.data_as(C.POINTER(C.c_uint64)), len(offs) - 1, nparts,
                                     cuts.ctypes.data_as(C.POINTER(C.c_uint64))))
    return [(int(cuts[i]), int(cuts[i + 1])) for i in range(nparts)]


def test_split_by_bytes_matches_python_rule(lib):
    rng = np.random.default_rng(11)
    cases = [np.full(1000, 150), rng.integers(0, 400, 5000), np.array([5]), np.array([0, 0, 0]),
             np.concatenate([np.full(300, 150), [2_000_000], np.full(300, 150)]),
             np.clip(np.exp(rng.normal(np.log(8000), 1.1, 3000)), 1000, 1_000_000).astype(np.int64),
             np.array([248956422, 242193529, 198295559, 190214555, 181538259, 170805979])]
    for lengths in cases:
        offs = np.zeros(len(lengths) + 1, np.uint64)
        np.cumsum(np.asarray(lengths, np.uint64), out=offs[1:])
        for nparts in (1, 2, 3, 4, 8):
            got = c_split_by_bytes(lib, offs, nparts)
            assert got == shard.split_by_bytes(offs, nparts), (len(lengths), nparts)
            assert got[0][0] == 0 and got[-1][1] == len(lengths) and all(a[1] == b[0] for a, b in zip(got, got[1:]))


def test_split_record_matches_python_rule(lib):
    for length in (0, 1, 15, 16, 17, 1000, 248956422, (1 << 33) + 5):
        for nparts in (1, 2, 3, 8):
            cuts = np.zeros(nparts + 1, np.uint64)
            lib.check(lib.skh_split_record(length, nparts, cuts.ctypes.data_as(C.POINTER(C.c_uint64))))
            got = [(int(cuts[i]), int(cuts[i + 1]), length - int(cuts[i + 1])) for i in range(nparts)]
            assert got == shard.split_record(length, nparts)
            assert all(int(c) % 16 == 0 for c in cuts[:-1])


def test_bad_arguments(lib):
    assert lib.skh_split_by_bytes(None, 0, 2, None) == -1
    cuts = np.zeros(2, np.uint64)
    assert lib.skh_split_record(10, 0, cuts.ctypes.data_as(C.POINTER(C.c_uint64))) == -1
