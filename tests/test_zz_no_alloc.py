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


def test_static_scratch_call_allocates_nothing_and_serves_any_stream_and_context(oracle_lib):
    """What the entry point exists for: free device memory is the same before and after its calls (warmed up
    once, so that the runtime's own one-time allocations are behind); a caller's stream works like the
    context's own; two contexts on one GPU share the one array, and calls that follow each other on one
    stream do not disturb each other."""
    import torch
    from seqkit_amd import load
    lib = load()
    ora = Oracle(oracle_lib)
    ctx_a, ctx_b = lib.init(0), lib.init(0)
    gap = (C.c_uint8 * 256)()
    for ch in b"- .":
        gap[ch] = 1
    rng = np.random.default_rng(6)
    try:
        batches = []
        for lengths in (np.concatenate([rng.integers(0, 300, 40000), [200000]]), rng.integers(1, 60, 900)):
            seq, _, offs = make_batch(lengths, seed=int(lengths.sum()) % 1000, letters=b"ACGT-. Nn")
            batches.append((seq, offs, to_dev(seq), to_dev(offs), torch.full((5,), -7, dtype=torch.int64, device="cuda:0")))
        stream = torch.cuda.Stream(device="cuda:0")
        sp = C.c_void_p(stream.cuda_stream)
        assert stream.cuda_stream != 0

        def call(ctx, k, st):
            seq, offs, d_seq, d_offs, d_out = batches[k]
            lib.check(lib.skh_stats_static_dev(ctx, ptr(d_seq), ptr(d_offs), len(offs) - 1, int(offs[-1]), gap, ptr(d_out), st))

        call(ctx_a, 0, sp)                       # warm-up: kernels loaded, the stream has its queue
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info(0)[0]
        # back to back on ONE caller stream, from two contexts of the same GPU: big, small, big, small
        call(ctx_a, 0, sp)
        call(ctx_b, 1, sp)
        stream.synchronize()
        got = [tuple(int(x) for x in batches[k][4].cpu()) for k in (0, 1)]
        call(ctx_b, 0, sp)
        call(ctx_a, 1, sp)
        stream.synchronize()
        got += [tuple(int(x) for x in batches[k][4].cpu()) for k in (0, 1)]
        free_after = torch.cuda.mem_get_info(0)[0]
        want = [ora.stats(batches[k][0], batches[k][1]) for k in (0, 1)]
        assert got == want + want
        assert free_after == free_before, (free_before, free_after)
    finally:
        lib.skh_destroy(ctx_a)
        lib.skh_destroy(ctx_b)
