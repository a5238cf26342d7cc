"""Planes of exactly skh_padded_size(total) bytes at the very end of a hipMalloc'd region.

The kernels load aligned 16-byte chunks and misaligned 16-byte windows around record ends; every
such load has to stay inside [plane, plane + skh_padded_size(total)).  torch's caching allocator
hands out blocks with slack behind them, which would hide a read just past a plane, so these cases
take their memory from hipMalloc directly (regions of whole 2-MiB pages, the plane ending where the
region ends) and use the shapes that put loads closest to the end: total % 16 != 0, a last record
shorter than 16 bytes, a short record before a long one and the reverse.  Outputs are compared with
the CPU oracle; the byte planes behind the outputs must stay untouched."""
import ctypes as C

import numpy as np
import pytest

from helpers import Oracle, make_batch

pytestmark = pytest.mark.gpu

PAGE = 2 << 20


class Region:
    """A hipMalloc'd region; `tail(n)` is the device address of its last n bytes."""

    def __init__(self, hip, nbytes):
        self.hip, self.size = hip, max((nbytes + PAGE - 1) // PAGE, 1) * PAGE
        self.base = C.c_void_p()
        assert hip.hipMalloc(C.byref(self.base), C.c_size_t(self.size)) == 0
        assert hip.hipMemset(self.base, 0xEE, C.c_size_t(self.size)) == 0

    def tail(self, n):
        return C.c_void_p(self.base.value + self.size - n)

    def put(self, n, host):
        assert self.hip.hipMemcpy(self.tail(n), host.ctypes.data_as(C.c_void_p), C.c_size_t(host.nbytes), 1) == 0

    def get(self, n):
        out = np.empty(n, np.uint8)
        assert self.hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), self.tail(n), C.c_size_t(n), 2) == 0
        return out

    def free(self):
        self.hip.hipFree(self.base)


SHAPES = {
    "short_reads_odd_total": np.concatenate([np.full(3000, 150), [7]]),
    "last_record_3_bytes": np.concatenate([np.full(999, 101), [3]]),
    "short_then_long": np.array([5, 70001, 9, 3, 150, 150, 40011, 2]),
    "long_then_short_tail": np.array([100003, 15, 1]),
    "one_long_odd": np.array([65537 + 11]),
    "single_byte": np.array([1]),
    "ont_like_odd": np.array([9001, 12345, 33000, 801, 17]),
}


@pytest.mark.parametrize("shape", sorted(SHAPES))
@pytest.mark.parametrize("kernels", ["auto", "span", "direct"])
def test_exact_size_planes(shape, kernels, oracle_lib, monkeypatch):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from seqkit_amd import load
    lib = load()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    ora = Oracle(oracle_lib)
    for k in ("SKH_SPAN_U", "SKH_STAGED"):
        monkeypatch.delenv(k, raising=False)
    if kernels == "span":
        monkeypatch.setenv("SKH_SPAN_U", "4")
    elif kernels == "direct":
        monkeypatch.setenv("SKH_SPAN_U", "0")
        monkeypatch.setenv("SKH_STAGED", "0")
    ctx = lib.init(0)
    lengths = SHAPES[shape]
    seq, qual, offs = make_batch(lengths, seed=len(lengths), letters=b"ACGTacgtNn-.RYKM", with_qual=True)
    nrec, total = len(offs) - 1, int(offs[-1])
    padded = int(lib.skh_padded_size(total))
    assert total % 16 != 0
    regs = [Region(hip, padded) for _ in range(4)]          # seq in, qual in, seq out, qual out
    r_off, r_cnt, r_st = Region(hip, 8 * (nrec + 1)), Region(hip, 8 * 4 * nrec), Region(hip, 40)
    try:
        regs[0].put(padded, seq[:padded]); regs[1].put(padded, qual[:padded])
        r_off.put(8 * (nrec + 1), offs)
        lut = ora.pair_table(1)
        lutp = lut.ctypes.data_as(C.POINTER(C.c_uint8))
        cls = np.frombuffer(bytes(lib.class_table(["G", "C", "A", "T"])), np.uint8).copy()
        gap = (C.c_uint8 * 256)()
        for ch in b"- .":
            gap[ch] = 1
        d_off = r_off.tail(8 * (nrec + 1))
        lib.check(lib.skh_revcomp_fastq_dev(ctx, regs[0].tail(padded), regs[2].tail(padded), regs[1].tail(padded),
                                            regs[3].tail(padded), d_off, nrec, total, lutp, 1, None))
        lib.check(lib.skh_count_dev(ctx, regs[0].tail(padded), d_off, nrec, total,
                                    cls.ctypes.data_as(C.POINTER(C.c_uint8)), 4, r_cnt.tail(32 * nrec), None))
        lib.check(lib.skh_stats_dev(ctx, regs[0].tail(padded), d_off, nrec, total, gap, r_st.tail(40), None))
        lib.check(lib.skh_sync(ctx))
        want_s, want_q = ora.transform(seq, qual, offs, lut, True)
        got_s, got_q = regs[2].get(padded), regs[3].get(padded)
        assert np.array_equal(got_s[:total], want_s[:total]) and np.array_equal(got_q[:total], want_q[:total])
        assert (got_s[total:] == 0xEE).all() and (got_q[total:] == 0xEE).all()
        # the bytes in front of the output planes were not written either
        for r in (regs[2], regs[3]):
            if r.size - padded >= 64:
                assert (r.get(padded + 64)[:64] == 0xEE).all()
        counts = r_cnt.get(32 * nrec).view(np.uint64).reshape(nrec, 4)
        assert np.array_equal(counts, ora.count(seq, offs, cls, 4))
        assert tuple(int(x) for x in r_st.get(40).view(np.int64)) == ora.stats(seq, offs)
    finally:
        lib.skh_destroy(ctx)
        for r in regs + [r_off, r_cnt, r_st]:
            r.free()
