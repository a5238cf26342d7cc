"""GPU parity of sequence validation (skh_validate_dev) against the oracle's ora_validate_seq,
applied record by record in file order as the reference's reader does: the first record with a
byte outside the alphabet within its first `limit` bytes, and that byte.  Pinned through the
oracle by the documented `seq -t dna` example (doc/docs/usage.md:322-324)."""
import ctypes as C
import zlib

import numpy as np
import pytest

from helpers import make_batch, ptr, to_dev

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(oracle_lib):
    import torch
    import seqkit_amd
    from seqkit_amd import capi
    lib = seqkit_amd.load()
    ctx = lib.init(0)
    oracle_lib.ora_validate_seq.restype = C.c_long
    oracle_lib.ora_validate_seq.argtypes = [C.POINTER(C.c_uint8), C.c_size_t, C.c_int, C.c_size_t]
    yield {"lib": lib, "ctx": ctx, "torch": torch, "capi": capi, "ora": oracle_lib}
    lib.skh_destroy(ctx)


def oracle_first_invalid(ora, seq, offs, alphabet, limit):
    for r in range(len(offs) - 1):
        a, b = int(offs[r]), int(offs[r + 1])
        rec = np.ascontiguousarray(seq[a:b])
        i = ora.ora_validate_seq(rec.ctypes.data_as(C.POINTER(C.c_uint8)), b - a, alphabet, limit)
        if i >= 0:
            return (r, int(i), int(rec[i]), 1)
    return (0, 0, 0, 0)


def gpu_first_invalid(env, seq, offs, alphabet, limit):
    torch, lib, capi = env["torch"], env["lib"], env["capi"]
    valid = (C.c_uint8 * 256)()
    lib.check(lib.skh_valid_table(alphabet, valid))
    d_seq, d_offs = to_dev(seq), to_dev(offs)
    d_res = torch.zeros(3, dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()
    lib.check(lib.skh_validate_dev(env["ctx"], ptr(d_seq), ptr(d_offs), len(offs) - 1, int(offs[-1]), valid, limit,
                                   ptr(d_res), None))
    lib.check(lib.skh_sync(env["ctx"]))
    r = capi.ValidateResult.from_buffer_copy(d_res.cpu().numpy().tobytes())
    return (r.record, r.offset, r.letter, r.found)


def plant(seq, offs, rec, off, byte):
    seq[int(offs[rec]) + off] = byte


LENGTHS = {
    "reads150": np.full(20000, 150),
    "ragged": None,
    "long": np.array([5, 40000, 0, 0, 12000, 3, 2_000_000, 17, 1500]),
    "tiny": np.array([0, 1, 0, 2, 1, 0, 0, 3]),
}


@pytest.mark.parametrize("case", sorted(LENGTHS))
@pytest.mark.parametrize("limit", [0, 1000, 10000])
def test_validate_matches_oracle(env, case, limit):
    rng = np.random.default_rng(zlib.crc32(("%s%d" % (case, limit)).encode()))
    lens = LENGTHS[case] if LENGTHS[case] is not None else rng.integers(0, 400, 30000)
    seq, _, offs = make_batch(lens, seed=11, letters=b"ACGTNacgtnRYKM-. ")
    nrec = len(lens)
    # all valid first
    for ab in (1, 3, 4, 5):
        assert gpu_first_invalid(env, seq, offs, ab, limit) == oracle_first_invalid(env["ora"], seq, offs, ab, limit)
    # plant invalid bytes at random places (several, so "first" matters), some beyond the limit
    for trial in range(6):
        s2 = seq.copy()
        nonempty = np.nonzero(np.asarray(lens) > 0)[0]
        for _ in range(int(rng.integers(1, 5))):
            r = int(rng.choice(nonempty))
            off = int(rng.integers(0, lens[r]))
            plant(s2, offs, r, off, int(rng.choice(np.frombuffer(b"eEUu*0@\x00\xff", np.uint8))))
        for ab in (1, 3):
            want = oracle_first_invalid(env["ora"], s2, offs, ab, limit)
            assert gpu_first_invalid(env, s2, offs, ab, limit) == want, (case, limit, trial, ab)
    assert nrec > 0


def test_validate_limit_edges(env):
    lens = np.array([3000, 3000, 3000])
    seq, _, offs = make_batch(lens, seed=2, letters=b"ACGT")
    for off, limit, found in ((999, 1000, 1), (1000, 1000, 0), (2999, 0, 1), (1500, 1000, 0)):
        s2 = seq.copy()
        plant(s2, offs, 1, off, ord("X"))
        got = gpu_first_invalid(env, s2, offs, 1, limit)
        assert got == oracle_first_invalid(env["ora"], s2, offs, 1, limit)
        assert got[3] == found
    # the last byte of the plane, and a later record after an ignored one
    s2 = seq.copy()
    plant(s2, offs, 0, 2000, ord("X")); plant(s2, offs, 2, 5, ord("Z"))
    assert gpu_first_invalid(env, s2, offs, 1, 1000) == (2, 5, ord("Z"), 1)
    s2 = seq.copy()
    plant(s2, offs, 2, 2999, ord("!"))
    assert gpu_first_invalid(env, s2, offs, 1, 0) == (2, 2999, ord("!"), 1)


def test_validate_empty_batch_and_unlimit(env):
    seq, _, offs = make_batch(np.array([0, 0]), seed=1, letters=b"ACGT")
    assert gpu_first_invalid(env, seq, offs, 1, 0) == (0, 0, 0, 0)
    seq, _, offs = make_batch(np.array([50, 60]), seed=1, letters=b"@#!?")
    assert gpu_first_invalid(env, seq, offs, 5, 0) == (0, 0, 0, 0)
    assert gpu_first_invalid(env, seq, offs, 0, 0)[3] == 1
