"""skh_stats_static_dev: the variant of skh_stats_dev whose kernel scratch lies in a static device
array (for callers that cannot allocate).  Its totals must equal skh_stats_dev's on two calls in a
row, the second on a smaller batch (the scratch keeps the first call's partial sums until the second
call clears them)."""
import ctypes as C

import numpy as np
import pytest

from helpers import Oracle, make_batch, ptr, to_dev

pytestmark = pytest.mark.gpu


def _totals(lib, fn, ctx, d_seq, d_offs, nrec, total, gap, stream=None):
    import torch
    d_out = torch.full((5,), -7, dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()
    lib.check(fn(ctx, ptr(d_seq), ptr(d_offs), nrec, total, gap, ptr(d_out), stream))
    lib.check(lib.skh_sync(ctx))
    torch.cuda.synchronize()
    return tuple(int(x) for x in d_out.cpu())


def test_static_scratch_stats_equals_allocating_entry_point(oracle_lib):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from seqkit_amd import load
    lib = load()
    ora = Oracle(oracle_lib)
    ctx = lib.init(0)
    gap = (C.c_uint8 * 256)()
    for ch in b"- .":
        gap[ch] = 1
    rng = np.random.default_rng(5)
    try:
        # gap letters everywhere, so that every workgroup leaves a partial sum in the scratch
        big = np.concatenate([rng.integers(0, 400, 60000), [300000, 0, 70001]])
        small = rng.integers(1, 90, 700)
        for lengths in (big, small):
            seq, _, offs = make_batch(lengths, seed=len(lengths), letters=b"ACGT-. Nn")
            nrec, total = len(offs) - 1, int(offs[-1])
            d_seq, d_offs = to_dev(seq), to_dev(offs)
            want = _totals(lib, lib.skh_stats_dev, ctx, d_seq, d_offs, nrec, total, gap)
            got = _totals(lib, lib.skh_stats_static_dev, ctx, d_seq, d_offs, nrec, total, gap)
            assert got == want == ora.stats(seq, offs)
        # beyond the stated maximum: a clear error, nothing launched
        rc = lib.skh_stats_static_dev(ctx, ptr(d_seq), ptr(d_offs), nrec, (1 << 40) + 16, gap, ptr(d_seq), None)
        assert rc == -5 and b"SKH_STATS_STATIC_MAX_BYTES" in lib.skh_last_error()
    finally:
        lib.skh_destroy(ctx)
