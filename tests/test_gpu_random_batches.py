"""Seeded random batches: every batch mixes runs of empty, tiny, short, medium and long records in random
order and proportion, so that wavefront steps and workgroup runs meet the kernels' rarer paths (steps that
store nothing, stretches larger than the stage, head chunks made byte by byte, long records next to empty
ones).  Reverse-complement (two planes and in place, with qualities), counts and stats totals must equal the
oracle's bit for bit."""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import Oracle, make_batch
from test_gpu_parity import SOFT, lib, run_count, run_revcomp, run_revcomp_inplace, run_stats  # noqa: F401  (lib is a fixture)

pytestmark = pytest.mark.gpu


def random_lengths(rng):
    parts = []
    for _ in range(int(rng.integers(3, 12))):
        kind = rng.integers(0, 7)
        n = int(rng.integers(1, 400))
        if kind == 0:
            parts.append(np.zeros(n, np.int64))                              # a run of empty records
        elif kind == 1:
            parts.append(rng.integers(0, 20, n))                             # tiny
        elif kind == 2:
            parts.append(np.full(n, int(rng.integers(30, 300))))             # fixed-length reads
        elif kind == 3:
            parts.append(rng.integers(20, 400, n))                           # ragged reads
        elif kind == 4:
            parts.append(rng.integers(400, 9000, max(1, n // 8)))            # medium
        elif kind == 5:
            parts.append(rng.integers(20000, 90000, max(1, n // 60)))        # around the long-record threshold
        else:
            parts.append(np.array([int(rng.integers(100_000, 1_500_000))]))  # one long record
    return np.concatenate(parts)


SETS = {"auto": {}, "staged": {"SKH_STAGED": "1", "SKH_SPAN_U": "0"}, "direct": {"SKH_STAGED": "0", "SKH_SPAN_U": "0"},
        "span": {"SKH_SPAN_U": "4"}, "no_hint": {"SKH_HINT": "0"}}


@pytest.fixture(scope="module", params=sorted(SETS))
def ctx0(lib, request):
    names = ("SKH_STAGED", "SKH_SPAN_U", "SKH_HINT")
    old = {k: os.environ.get(k) for k in names}
    for k in names:
        os.environ.pop(k, None)
    os.environ.update(SETS[request.param])
    c = lib.init(0)
    for k in names:
        os.environ.pop(k, None)
        if old[k] is not None:
            os.environ[k] = old[k]
    yield c
    lib.skh_destroy(c)


@pytest.mark.parametrize("seed", range(40))
def test_random_batch(lib, ctx0, oracle_lib, seed):
    ora = Oracle(oracle_lib)
    rng = np.random.default_rng(1000 + seed)
    lengths = random_lengths(rng)
    seq, qual, offs = make_batch(lengths, seed=seed, letters=SOFT, with_qual=True)
    total = int(offs[-1])
    lut = ora.pair_table(1)
    want_s, want_q = ora.transform(seq, qual, offs, lut, True)
    got_s, got_q = run_revcomp(lib, ctx0, seq, qual, offs, lut, True)
    assert np.array_equal(got_s[:total], want_s[:total]) and np.array_equal(got_q[:total], want_q[:total])
    got_s, got_q = run_revcomp_inplace(lib, ctx0, seq, qual, offs, lut, True)
    assert np.array_equal(got_s[:total], want_s[:total]) and np.array_equal(got_q[:total], want_q[:total])
    assert np.array_equal(got_s[total:], seq[total:])
    cls = np.frombuffer(bytes(lib.class_table(["G", "C", "A", "T", "N-"])), np.uint8)
    assert np.array_equal(run_count(lib, ctx0, seq, offs, cls, 5), ora.count(seq, offs, cls, 5))
    assert run_stats(lib, ctx0, seq, offs) == ora.stats(seq, offs)
