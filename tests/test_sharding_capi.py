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


def test_split_plane_matches_python_rule_and_tiles_the_plane(lib):
    rng = np.random.default_rng(12)
    cases = [np.full(1000, 150), rng.integers(0, 400, 5000), np.array([5]), np.array([0, 0, 0]), np.array([5_000_003]),
             np.concatenate([np.full(300, 150), [2_000_000], np.full(300, 150)]),
             np.concatenate([np.zeros(40, np.int64), [100_000], np.zeros(40, np.int64)]),
             np.clip(np.exp(rng.normal(np.log(8000), 1.1, 3000)), 1000, 1_000_000).astype(np.int64),
             np.array([248956422, 242193529, 198295559, 190214555, 181538259, 170805979])]
    for lengths in cases:
        offs = np.zeros(len(lengths) + 1, np.uint64)
        np.cumsum(np.asarray(lengths, np.uint64), out=offs[1:])
        total = int(offs[-1])
        for nparts in (1, 2, 3, 4, 8, 24):
            cuts = np.zeros(nparts + 1, np.uint64)
            lib.check(lib.skh_split_plane(offs.ctypes.data_as(C.POINTER(C.c_uint64)), len(offs) - 1, nparts,
                                          cuts.ctypes.data_as(C.POINTER(C.c_uint64))))
            got = [int(c) for c in cuts]
            assert got == shard.split_plane(offs, nparts), (len(lengths), nparts)
            assert got[0] == 0 and got[-1] == total and all(a <= b for a, b in zip(got, got[1:]))
            # byte-balanced: no share is more than an eighth of a share (and a chunk) off
            share = total // nparts
            assert all(abs((b - a) - share) <= share // 8 + 32 for a, b in zip(got, got[1:])), (len(lengths), nparts)
            # a cut inside a record lies at a multiple of 16 bytes of that record; pieces tile every record once
            seen = {}
            for a, b in zip(got, got[1:]):
                for r, x, y in shard.pieces_of_share(offs, a, b):
                    assert x % 16 == 0 and seen.get(r, 0) == x
                    seen[r] = y
            assert all(seen.get(r, 0) == int(l) for r, l in enumerate(lengths))


def test_bad_arguments(lib):
    assert lib.skh_split_by_bytes(None, 0, 2, None) == -1
    cuts = np.zeros(2, np.uint64)
    assert lib.skh_split_record(10, 0, cuts.ctypes.data_as(C.POINTER(C.c_uint64))) == -1
