"""One pinned batch cut over several contexts (skh_*_multi): every result must equal the oracle's on
the whole batch, whatever the cuts -- at record boundaries, inside long records, through runs of empty
records.  The test box has one GPU, so the contexts share it (they still have their own worker
threads, streams and device planes); with two or more GPUs visible distinct devices are used as well."""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import Oracle, make_batch

pytestmark = pytest.mark.gpu

u8p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint64)


def device_lists():
    import torch
    lists = [[0], [0, 0], [0, 0, 0], [0] * 5]
    if torch.cuda.device_count() >= 2:
        lists += [[0, 1], [1, 0, 1]]
    return lists


def length_cases():
    rng = np.random.default_rng(11)
    return {
        "one_5mb_record": np.array([5_000_003]),
        "mixed_long": np.concatenate([rng.integers(0, 400, 3000), [1_200_000, 0, 17, 700_001], rng.integers(50, 300, 2000)]),
        "reads_150": np.full(40_000, 150),
        "ragged": rng.integers(0, 2000, 5000),
        "empty_runs": np.concatenate([np.zeros(500, np.int64), [40], np.zeros(300, np.int64), [100_000], np.zeros(100, np.int64)]),
        "all_empty": np.zeros(1000, np.int64),
        "two_long": np.array([900_001, 2_000_017]),
        "tiny": np.array([3, 0, 1]),
    }


class Batch:
    def __init__(self, lib, ctx, seq, qual, offs):
        self.lib = lib
        nrec, total = len(offs) - 1, int(offs[-1])
        self.h = C.c_void_p()
        lib.check(lib.skh_batch_alloc(ctx, max(total, 16), max(nrec, 1), 1 if qual is not None else 0, C.byref(self.h)))
        self.nrec, self.total = nrec, total
        self.seq = np.ctypeslib.as_array(C.cast(lib.skh_batch_seq(self.h), u8p), shape=(lib.skh_batch_max_bytes(self.h),))
        self.offs = np.ctypeslib.as_array(C.cast(lib.skh_batch_offsets(self.h), u64p), shape=(nrec + 1,))
        self.qual = None
        if qual is not None:
            self.qual = np.ctypeslib.as_array(C.cast(lib.skh_batch_qual(self.h), u8p), shape=(lib.skh_batch_max_bytes(self.h),))
        self.fill(seq, qual, offs)

    def fill(self, seq, qual, offs):
        self.seq[:self.total] = seq[:self.total]
        if qual is not None:
            self.qual[:self.total] = qual[:self.total]
        self.offs[:] = offs

    def free(self):
        self.lib.skh_batch_free(self.h)


@pytest.fixture(scope="module")
def env():
    from seqkit_amd import load
    return load()


def make_multi(lib, devices, chunk=None):
    old = os.environ.get("SKH_MULTI_CHUNK")
    if chunk:
        os.environ["SKH_MULTI_CHUNK"] = str(chunk)
    ctxs = [lib.init(d) for d in devices]
    arr = (C.c_void_p * len(ctxs))(*[c.value for c in ctxs])
    m = C.c_void_p()
    try:
        lib.check(lib.skh_multi_init(arr, len(ctxs), C.byref(m)))
    finally:
        if chunk:
            if old is None:
                del os.environ["SKH_MULTI_CHUNK"]
            else:
                os.environ["SKH_MULTI_CHUNK"] = old
    return ctxs, m


def shares_of(lib, m):
    from seqkit_amd.capi import ShareInfo
    n = lib.skh_multi_size(m)
    info = (ShareInfo * n)()
    lib.check(lib.skh_multi_shares(m, info))
    return list(info)


@pytest.mark.parametrize("chunk", [None, 4096, 100_000])
@pytest.mark.parametrize("devices", device_lists(), ids=lambda d: "dev" + "".join(map(str, d)))
def test_multi_matches_oracle(env, oracle_lib, devices, chunk):
    lib, ora = env, Oracle(oracle_lib)
    ctxs, m = make_multi(lib, devices, chunk)
    lut = lib.pair_table(1)
    lut_np = np.frombuffer(bytes(lut), np.uint8)
    cls4, cls8 = lib.class_table(["G", "C", "A", "T"]), lib.class_table(["G", "C", "A", "T", "N", "-", ".", "RY"])
    gap = (C.c_uint8 * 256)()
    for ch in b"- .":
        gap[ch] = 1
    try:
        for name, lengths in length_cases().items():
            seq, qual, offs = make_batch(lengths, seed=5, letters=b"ACGTacgtNnRY-. ", with_qual=True)
            nrec, total = len(offs) - 1, int(offs[-1])
            b = Batch(lib, ctxs[0], seq, qual, offs)
            try:
                # reverse-complement with qualities, reverse only, complement only
                for use_lut, rev, rq in ((True, 1, 1), (False, 1, 0), (True, 0, 1)):
                    b.fill(seq, qual, offs)
                    lib.check(lib.skh_revcomp_multi(m, b.h, nrec, lut if use_lut else None, rev, rq))
                    want_s, want_q = ora.transform(seq, qual if (rev and rq) else None, offs, lut_np if use_lut else None, rev)
                    assert np.array_equal(b.seq[:total], want_s[:total]), (name, use_lut, rev)
                    assert np.array_equal(b.qual[:total], (want_q if (rev and rq) else qual)[:total]), (name, "qual", rev, rq)
                    assert np.array_equal(b.offs, offs)
                b.fill(seq, qual, offs)
                for cls, nc in ((cls4, 4), (cls8, 8)):
                    counts = np.full(max(nrec, 1) * nc, 0xABCD, np.uint64)
                    lib.check(lib.skh_count_multi(m, b.h, nrec, cls, nc, counts.ctypes.data_as(u64p)))
                    want = ora.count(seq, offs, np.frombuffer(bytes(cls), np.uint8), nc)
                    assert np.array_equal(counts[:nrec * nc].reshape(nrec, nc), want), (name, nc)
                from seqkit_amd import StatsPartial
                tot = StatsPartial()
                lib.skh_stats_init(C.byref(tot))
                lib.check(lib.skh_stats_multi(m, b.h, nrec, gap, C.byref(tot)))
                assert tot.as_tuple() == ora.stats(seq, offs), name
                sh = shares_of(lib, m)
                assert sh[0].first_byte == 0 and max(s.last_byte for s in sh) == total
                assert sum(s.last_byte - s.first_byte for s in sh) == total
            finally:
                b.free()
    finally:
        lib.skh_multi_destroy(m)
        for c in ctxs:
            lib.skh_destroy(c)


def test_one_record_is_worked_on_by_every_context(env, oracle_lib):
    """A single 5 MB record over three contexts: each of them gets a third of the record (cut inside it, at
    multiples of 16 bytes), and the three reversed pieces land mirrored in the one output."""
    lib, ora = env, Oracle(oracle_lib)
    ctxs, m = make_multi(lib, [0, 0, 0])
    try:
        seq, _, offs = make_batch([5_000_003], seed=9, letters=b"ACGTN")
        b = Batch(lib, ctxs[1], seq, None, offs)          # a batch of any context will do
        lut = lib.pair_table(1)
        lib.check(lib.skh_revcomp_multi(m, b.h, 1, lut, 1, 0))
        want, _ = ora.transform(seq, None, offs, np.frombuffer(bytes(lut), np.uint8), 1)
        assert np.array_equal(b.seq[:5_000_003], want[:5_000_003])
        sh = shares_of(lib, m)
        assert len(sh) == 3
        for k, s in enumerate(sh):
            assert s.first_record == 0 and s.nrecords == 1
            assert s.last_byte - s.first_byte > 1_600_000            # about a third each
            assert s.first_byte % 16 == 0 and (s.head_cut > 0) == (k > 0) and (s.tail_cut > 0) == (k < 2)
            assert s.seconds > 0
        b.free()
    finally:
        lib.skh_multi_destroy(m)
        for c in ctxs:
            lib.skh_destroy(c)


def test_multi_at_scale_round_trip(env):
    """24 records with chromosome-like lengths (1/20 scale) and 2 M reads, over four contexts: twice the
    reverse-complement is the identity, counts of the result are the counts of the input with the
    classes of complementary letters swapped, stats totals are those of the offsets."""
    lib = env
    ctxs, m = make_multi(lib, [0, 0, 0, 0])
    lut = lib.pair_table(1)
    cls = lib.class_table(["G", "C", "A", "T"])
    gap = (C.c_uint8 * 256)()
    gap[ord("-")] = 1
    try:
        chrom = np.array([248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636,
                          138394717, 133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345,
                          83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415]) // 20
        for lengths in (chrom, np.full(2_000_000, 150)):
            seq, qual, offs = make_batch(lengths, seed=21, letters=b"ACGTN-", with_qual=True)
            nrec, total = len(offs) - 1, int(offs[-1])
            b = Batch(lib, ctxs[0], seq, qual, offs)
            try:
                c0 = np.zeros(nrec * 4, np.uint64)
                lib.check(lib.skh_count_multi(m, b.h, nrec, cls, 4, c0.ctypes.data_as(u64p)))
                lib.check(lib.skh_revcomp_multi(m, b.h, nrec, lut, 1, 1))
                assert not np.array_equal(b.seq[:total], seq[:total])
                c1 = np.zeros(nrec * 4, np.uint64)
                lib.check(lib.skh_count_multi(m, b.h, nrec, cls, 4, c1.ctypes.data_as(u64p)))
                assert np.array_equal(c0.reshape(nrec, 4)[:, [1, 0, 3, 2]], c1.reshape(nrec, 4))
                assert int(c0.sum()) == int(np.isin(seq[:total], np.frombuffer(b"ACGT", np.uint8)).sum())
                lib.check(lib.skh_revcomp_multi(m, b.h, nrec, lut, 1, 1))
                assert np.array_equal(b.seq[:total], seq[:total]) and np.array_equal(b.qual[:total], qual[:total])
                from seqkit_amd import StatsPartial
                tot = StatsPartial()
                lib.skh_stats_init(C.byref(tot))
                lib.check(lib.skh_stats_multi(m, b.h, nrec, gap, C.byref(tot)))
                assert tot.as_tuple() == (nrec, total, int((seq[:total] == ord("-")).sum()), int(lengths.min()), int(lengths.max()))
                sh = shares_of(lib, m)
                sizes = [s.last_byte - s.first_byte for s in sh]
                assert max(sizes) - min(sizes) <= total // 4 // 8 + 16, sizes     # byte-balanced
            finally:
                b.free()
    finally:
        lib.skh_multi_destroy(m)
        for c in ctxs:
            lib.skh_destroy(c)
