"""The LDS-staged reverse kernel waits for its global->LDS loads with a hand-counted s_waitcnt.  The checked
build of the library (-DSKH_STAGE_CHECK, libseqkit_hip_checked.so) compares every stage buffer and offsets
slice with global memory right after that wait and makes the entry point fail when one differs; here it runs
over the shapes that take the kernel's different paths, with one to six steps per wavefront."""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import Oracle, dev_empty_like_plane, make_batch, ptr, to_dev

pytestmark = pytest.mark.gpu

CHECKED = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "seqkit_amd",
                       "libseqkit_hip_checked.so")


@pytest.fixture(scope="module")
def checked():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from seqkit_amd.capi import Lib
    return Lib(CHECKED)


def lengths_of(shape):
    rng = np.random.default_rng(3)
    return {
        "reads150": np.full(400_000, 150),
        "reads100": np.full(300_000, 100),
        "reads250": np.full(200_000, 250),
        "trimmed": rng.integers(30, 152, 300_000),
        "ragged0_300": rng.integers(0, 301, 200_000),            # overflowing stretches, empty records
        "tiny": rng.integers(0, 21, 400_000),
        "empty_runs": np.concatenate([np.zeros(4096, np.int64), np.full(50_000, 150), np.zeros(6400, np.int64),
                                      np.ones(4000, np.int64), np.full(100_000, 150), np.zeros(30_000, np.int64), [7]]),
        "with_long": np.concatenate([np.full(50_000, 150), [100_000, 3, 1 << 20], np.full(50_000, 150)]),
    }[shape]


@pytest.mark.parametrize("steps", ["1", "2", "6"])
@pytest.mark.parametrize("shape", ["reads150", "reads100", "reads250", "trimmed", "ragged0_300", "tiny", "empty_runs", "with_long"])
def test_stage_buffers_complete(checked, oracle_lib, shape, steps, monkeypatch):
    import torch
    monkeypatch.setenv("SKH_STAGED", "1")
    monkeypatch.setenv("SKH_SPAN_U", "0")
    monkeypatch.setenv("SKH_STAGED_STEPS", steps)
    ctx = checked.init(0)
    try:
        seq, _, offs = make_batch(lengths_of(shape), seed=11, letters=b"ACGTacgtN")
        total = int(offs[-1])
        ora = Oracle(oracle_lib)
        lut = ora.pair_table(1)
        want, _ = ora.transform(seq, None, offs, lut, True)
        d_in, d_offs, d_out = to_dev(seq), to_dev(offs), dev_empty_like_plane(len(seq))
        lutp = np.ascontiguousarray(lut, np.uint8).ctypes.data_as(C.POINTER(C.c_uint8))
        torch.cuda.synchronize()
        for _ in range(3):      # the entry point fails if any wavefront step met an incomplete stage buffer
            checked.check(checked.skh_revcomp_dev(ctx, ptr(d_in), ptr(d_out), ptr(d_offs), len(offs) - 1, total, lutp, 1, None))
        checked.check(checked.skh_sync(ctx))
        assert np.array_equal(d_out.cpu().numpy()[:total], want[:total])
    finally:
        checked.skh_destroy(ctx)
