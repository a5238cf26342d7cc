"""GPU parity: every device entry point of libseqkit_hip.so, called through the C ABI,
must reproduce the CPU oracle bit for bit on the same seeded inputs."""
import ctypes as C
import os
import zlib

import numpy as np
import pytest

from helpers import GOLDEN, Oracle, dev_empty_like_plane, make_batch, ptr, read_fastx_plain, to_dev

pytestmark = pytest.mark.gpu

SOFT = b"ACGTacgtNnRYKMSWBDHVrykmswbdhv-. *Uu"


@pytest.fixture(scope="module")
def lib():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from seqkit_amd import load
    return load()


# every reverse implementation must give the same bytes; the variables are read by skh_init
KERNEL_SETS = {
    "staged": {"SKH_SPAN_U": "0", "SKH_STAGED": "1"},                # LDS-staged lane groups (default for reads of 128-191 bases)
    "flat": {"SKH_SPAN_U": "0", "SKH_STAGED": "2"},                  # pieces of all records of a step dealt over the lanes, rows
                                                                     # put together in LDS (default below 3000 bases otherwise)
    "span": {"SKH_SPAN_U": "4"},                                     # lanes own consecutive output chunks (on request only)
    "direct": {"SKH_SPAN_U": "0", "SKH_STAGED": "0"},                # lane groups without LDS staging (default from 300 bases on)
    "fused": {"SKH_SPAN_U": "0", "SKH_STAGED": "0", "SKH_FUSE_FASTQ": "1"},   # bases and qualities in one launch
    "auto": {},                                                      # kernel chosen from the mean record length
}


@pytest.fixture(scope="module", params=sorted(KERNEL_SETS))
def ctx(lib, request):
    env = KERNEL_SETS[request.param]
    names = ("SKH_SPAN_U", "SKH_STAGED", "SKH_FUSE_FASTQ")
    old = {k: os.environ.get(k) for k in names}
    for k in names:
        os.environ.pop(k, None)
    os.environ.update(env)
    c = lib.init(0)
    for k in names:
        os.environ.pop(k, None)
        if old[k] is not None:
            os.environ[k] = old[k]
    yield c
    lib.skh_destroy(c)


@pytest.fixture(scope="module")
def ora(oracle_lib):
    return Oracle(oracle_lib)


def length_cases():
    rng = np.random.default_rng(7)
    return {
        "illumina150": np.full(5000, 150),
        "ragged0_300": rng.integers(0, 301, 4000),
        "tiny0_20": rng.integers(0, 21, 6000),
        "contigs": np.array([6, 2, 4, 10, 5, 3, 8, 7, 9]),
        "single1": np.array([1]),
        "single16": np.array([16]),
        "empties": np.array([0, 0, 5, 0, 0, 0, 33, 0]),
        "medium1k_20k": rng.integers(1000, 20001, 300),
        "around_threshold": np.array([32767, 32768, 32769, 15, 32783, 1, 65536, 65537, 0, 40000]),
        "mixed_long": np.concatenate([rng.integers(1, 400, 500), [100000, 3, 1 << 20, 70, 0, 250001],
                                      rng.integers(1, 400, 300)]),
        "one_5mb": np.array([5 * 1024 * 1024 + 13]),
        "two_long_unaligned": np.array([7, 300007, 11, 299993, 5]),
        # short reads whose mean length a few records of megabases inflate: the short kernels take their
        # lanes per record from the sampled short records, the long-record kernels walk the offsets
        "reads_plus_contigs": np.concatenate([np.full(1500, 150), [2_000_003], np.full(1500, 150), [1_500_000, 151],
                                              rng.integers(30, 152, 2000), [40000]]),
        # whole wavefront steps of empty or one-byte records between reads: steps that store nothing
        "empty_runs": np.concatenate([np.zeros(100, np.int64), np.full(50, 150), np.zeros(64, np.int64), np.ones(40, np.int64),
                                      np.full(100, 150), np.zeros(300, np.int64), [7], np.zeros(16, np.int64)]),
        "long_run_of_longs": np.concatenate([rng.integers(33000, 70000, 40), [12], rng.integers(100, 200, 600)]),
    }


CASES = length_cases()


def run_revcomp(lib, ctx, seq, qual, offs, lut, do_reverse):
    import torch
    total = int(offs[-1])
    d_in, d_offs = to_dev(seq), to_dev(offs)
    d_out = dev_empty_like_plane(len(seq))
    lutp = np.ascontiguousarray(lut, np.uint8).ctypes.data_as(C.POINTER(C.c_uint8)) if lut is not None else None
    d_qin = to_dev(qual) if qual is not None else None
    d_qout = dev_empty_like_plane(len(seq)) if qual is not None else None
    torch.cuda.synchronize()        # torch filled the buffers on its own stream
    if qual is None:
        lib.check(lib.skh_revcomp_dev(ctx, ptr(d_in), ptr(d_out), ptr(d_offs), len(offs) - 1, total, lutp,
                                      int(do_reverse), None))
        lib.check(lib.skh_sync(ctx))
        torch.cuda.synchronize()
        return d_out.cpu().numpy(), None
    lib.check(lib.skh_revcomp_fastq_dev(ctx, ptr(d_in), ptr(d_out), ptr(d_qin), ptr(d_qout), ptr(d_offs),
                                        len(offs) - 1, total, lutp, int(do_reverse), None))
    lib.check(lib.skh_sync(ctx))
    torch.cuda.synchronize()
    return d_out.cpu().numpy(), d_qout.cpu().numpy()


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("mode", ["revcomp", "reverse", "complement", "fastq"])
def test_transform_matches_oracle(lib, ctx, ora, case, mode):
    lengths = CASES[case]
    fastq = mode == "fastq"
    seq, qual, offs = make_batch(lengths, seed=zlib.crc32(case.encode()) % 1000, letters=SOFT, with_qual=fastq)
    total = int(offs[-1])
    lut = ora.pair_table(1) if mode != "reverse" else None
    do_reverse = mode != "complement"
    want_s, want_q = ora.transform(seq, qual, offs, lut, do_reverse)
    got_s, got_q = run_revcomp(lib, ctx, seq, qual, offs, lut, do_reverse)
    assert np.array_equal(got_s[:total], want_s[:total])
    # bytes past the batch are never written
    assert (got_s[total:] == 0xEE).all()
    if fastq:
        assert np.array_equal(got_q[:total], want_q[:total])
        assert (got_q[total:] == 0xEE).all()


def run_revcomp_inplace(lib, ctx, seq, qual, offs, lut, do_reverse):
    import torch
    total = int(offs[-1])
    d_seq, d_offs = to_dev(seq), to_dev(offs)
    d_qual = to_dev(qual) if qual is not None else None
    lutp = np.ascontiguousarray(lut, np.uint8).ctypes.data_as(C.POINTER(C.c_uint8)) if lut is not None else None
    torch.cuda.synchronize()
    lib.check(lib.skh_revcomp_inplace_dev(ctx, ptr(d_seq), ptr(d_qual) if qual is not None else None, ptr(d_offs),
                                          len(offs) - 1, total, lutp, int(do_reverse), None))
    lib.check(lib.skh_sync(ctx))
    return d_seq.cpu().numpy(), d_qual.cpu().numpy() if qual is not None else None


# in place: the LDS-staged kernel with one plane (short reads), window pairs per lane, or the flat kernel; all for every case
@pytest.fixture(scope="module", params=["auto", "staged", "pairs", "flat"])
def ctx_default(lib, request):
    old = os.environ.get("SKH_STAGED")
    if request.param != "auto":
        os.environ["SKH_STAGED"] = {"staged": "1", "pairs": "0", "flat": "2"}[request.param]
    c = lib.init(0)
    os.environ.pop("SKH_STAGED", None)
    if old is not None:
        os.environ["SKH_STAGED"] = old
    yield c
    lib.skh_destroy(c)


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("mode", ["revcomp", "reverse", "complement", "fastq", "fastq_complement"])
def test_inplace_matches_oracle(lib, ctx_default, ora, case, mode):
    # one plane per kind, overwritten: (*Seq).ReverseInplace / ComplementInplace as the reference applies them
    lengths = CASES[case]
    fastq = mode.startswith("fastq")
    seq, qual, offs = make_batch(lengths, seed=zlib.crc32(case.encode()) % 1000 + 1, letters=SOFT, with_qual=fastq)
    total = int(offs[-1])
    lut = ora.pair_table(1) if mode != "reverse" else None
    do_reverse = mode not in ("complement", "fastq_complement")
    want_s, want_q = ora.transform(seq, qual, offs, lut, do_reverse)
    got_s, got_q = run_revcomp_inplace(lib, ctx_default, seq, qual, offs, lut, do_reverse)
    assert np.array_equal(got_s[:total], want_s[:total])
    assert np.array_equal(got_s[total:], seq[total:])            # the padding is not touched
    if fastq:
        assert np.array_equal(got_q[:total], want_q[:total])
        assert np.array_equal(got_q[total:], qual[total:])


def test_inplace_twice_is_identity(lib, ctx_default, ora):
    lengths = np.concatenate([CASES["ragged0_300"], CASES["mixed_long"], CASES["tiny0_20"][:500]])
    seq, qual, offs = make_batch(lengths, seed=5, letters=b"ACGTacgtN", with_qual=True)
    lut = ora.pair_table(1)
    once_s, once_q = run_revcomp_inplace(lib, ctx_default, seq, qual, offs, lut, True)
    twice_s, twice_q = run_revcomp_inplace(lib, ctx_default, once_s, once_q, offs, lut, True)
    assert np.array_equal(twice_s, seq) and np.array_equal(twice_q, qual)


@pytest.mark.parametrize("count_u", ["3", "4", "5"])
@pytest.mark.parametrize("logg", range(7))
def test_every_group_width(lib, ora, logg, count_u, monkeypatch):
    # the lanes-per-record choice and the chunks a lane of the counting kernel takes per round are speed knobs:
    # every combination must give the same bytes
    monkeypatch.setenv("SKH_LOGG", str(logg))
    monkeypatch.setenv("SKH_SPAN_U", "0")
    monkeypatch.setenv("SKH_COUNT_U", count_u)
    c = lib.init(0)
    try:
        lengths = np.concatenate([CASES["ragged0_300"][:700], [40000, 0, 33000, 17], CASES["tiny0_20"][:300],
                                  CASES["medium1k_20k"][:20]])
        seq, qual, offs = make_batch(lengths, seed=logg, letters=SOFT, with_qual=True)
        total = int(offs[-1])
        lut = ora.pair_table(3)
        want_s, want_q = ora.transform(seq, qual, offs, lut, True)
        got_s, got_q = run_revcomp(lib, c, seq, qual, offs, lut, True)
        assert np.array_equal(got_s[:total], want_s[:total]) and np.array_equal(got_q[:total], want_q[:total])
        got_s, got_q = run_revcomp_inplace(lib, c, seq, qual, offs, lut, True)
        assert np.array_equal(got_s[:total], want_s[:total]) and np.array_equal(got_q[:total], want_q[:total])
        cls = np.frombuffer(bytes(lib.class_table(["A", "C", "G", "T", "N-"])), np.uint8)
        assert np.array_equal(run_count(lib, c, seq, offs, cls, 5), ora.count(seq, offs, cls, 5))
    finally:
        lib.skh_destroy(c)


def test_flat_kernel_step_shapes(lib, ora, monkeypatch):
    # The flat reverse kernel takes as many records per step as fit its 7-KiB LDS stage by the batch's MEAN length:
    # batches whose steps are even, ragged (slot table in LDS), hold records far beyond twice the mean (copied
    # after the slots), do not fit the stage (written record by record) or hold records beyond 32 KiB (whose last
    # partial chunk this kernel writes, the rest the long-record kernel), with empty and tiny records in between.
    monkeypatch.setenv("SKH_STAGED", "2")
    monkeypatch.setenv("SKH_SPAN_U", "0")
    rng = np.random.default_rng(5)
    batches = {
        "even150": np.full(3000, 150),
        "ragged30_151": rng.integers(30, 152, 5000),
        "heavy_tail": np.clip(np.exp(rng.normal(np.log(100), 0.6, 5000)), 1, 6000).astype(np.int64),
        "one_long_per_step": np.where(np.arange(4000) % 37 == 0, 2500, rng.integers(0, 90, 4000)),
        "overflowing_steps": np.where(np.arange(3000) % 50 < 5, 7000, 40),
        "around_the_stage": np.array([7167, 7168, 7169, 7100, 15, 16, 17, 0, 7184, 1, 3584, 3584, 10000, 31, 32, 33] * 20),
        "long_records_between": np.concatenate([rng.integers(0, 300, 500), [40000, 0, 33000, 17, 32768, 32769, 5, 100001, 3], rng.integers(0, 40, 300)]),
        "kilobases": rng.integers(800, 2400, 400),
        "single": np.array([150]),
        "all_empty": np.zeros(100, np.int64),
    }
    for steps in ("1", "3"):                           # steps per wavefront
        monkeypatch.setenv("SKH_STAGED_STEPS", steps)
        c = lib.init(0)
        try:
            for name, lengths in batches.items():
                seq, qual, offs = make_batch(np.asarray(lengths, np.int64), seed=len(name), letters=SOFT, with_qual=True)
                total = int(offs[-1])
                lut = ora.pair_table(3)
                want_s, want_q = ora.transform(seq, qual, offs, lut, True)
                got_s, got_q = run_revcomp(lib, c, seq, qual, offs, lut, True)
                assert np.array_equal(got_s[:total], want_s[:total]), (name, steps)
                assert np.array_equal(got_q[:total], want_q[:total]), (name, steps)
                got_s, got_q = run_revcomp_inplace(lib, c, seq, qual, offs, lut, True)
                assert np.array_equal(got_s[:total], want_s[:total]), (name, steps, "in place")
                assert np.array_equal(got_q[:total], want_q[:total]), (name, steps, "in place")
        finally:
            lib.skh_destroy(c)


def run_count(lib, ctx, seq, offs, class_of, nclass):
    import torch
    nrec, total = len(offs) - 1, int(offs[-1])
    d_seq, d_offs = to_dev(seq), to_dev(offs)
    d_counts = torch.full((max(nrec * nclass, 1),), -1, dtype=torch.int64, device="cuda:0")
    clsp = np.ascontiguousarray(class_of, np.uint8).ctypes.data_as(C.POINTER(C.c_uint8))
    torch.cuda.synchronize()
    lib.check(lib.skh_count_dev(ctx, ptr(d_seq), ptr(d_offs), nrec, total, clsp, nclass, ptr(d_counts), None))
    lib.check(lib.skh_sync(ctx))
    return d_counts.cpu().numpy().view(np.uint64)[:nrec * nclass].reshape(nrec, nclass)


def class_tables(lib):
    t = {}
    t["gc_at(4)"] = (np.frombuffer(bytes(lib.class_table(["G", "C", "A", "T"])), np.uint8), 4)
    t["gaps(1)"] = (np.frombuffer(bytes(lib.class_table(["- ."])), np.uint8), 1)
    t["acgtn_gap(6)"] = (np.frombuffer(bytes(lib.class_table(["A", "C", "G", "T", "N", "-. "])), np.uint8), 6)
    t["eight(8)"] = (np.frombuffer(bytes(lib.class_table(["A", "C", "G", "T", "N", "R", "Y", "-"])), np.uint8), 8)
    t["all_bytes_classified(3)"] = ((np.arange(256) % 3).astype(np.uint8), 3)   # no neutral byte exists
    t["all_bytes_classified(7)"] = ((np.arange(256) % 7).astype(np.uint8), 7)
    return t


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("table", ["gc_at(4)", "gaps(1)", "acgtn_gap(6)", "eight(8)",
                                   "all_bytes_classified(3)", "all_bytes_classified(7)"])
def test_count_matches_oracle(lib, ctx, ora, case, table):
    cls, nclass = class_tables(lib)[table]
    seq, _, offs = make_batch(CASES[case], seed=len(case), letters=SOFT)
    assert np.array_equal(run_count(lib, ctx, seq, offs, cls, nclass), ora.count(seq, offs, cls, nclass))


def test_count_saturated_lanes(lib, ctx, ora):
    # one class only and long runs: every 8-bit partial counter reaches its flush point
    cls = np.frombuffer(bytes(lib.class_table(["A"])), np.uint8)
    seq, _, offs = make_batch([32768, 4000, 32768, 240, 241, 255, 256, 100000], seed=1, letters=b"A")
    assert np.array_equal(run_count(lib, ctx, seq, offs, cls, 1), ora.count(seq, offs, cls, 1))


def run_stats(lib, ctx, seq, offs, gaps=b"- ."):
    import torch
    from seqkit_amd import StatsPartial
    is_gap = np.zeros(256, np.uint8)
    is_gap[list(gaps)] = 1
    d_seq, d_offs = to_dev(seq), to_dev(offs)
    d_out = torch.zeros(5, dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()
    lib.check(lib.skh_stats_dev(ctx, ptr(d_seq), ptr(d_offs), len(offs) - 1, int(offs[-1]),
                                is_gap.ctypes.data_as(C.POINTER(C.c_uint8)), ptr(d_out), None))
    lib.check(lib.skh_sync(ctx))
    return tuple(int(x) for x in d_out.cpu().numpy())


@pytest.mark.parametrize("case", sorted(CASES))
def test_stats_matches_oracle(lib, ctx, ora, case):
    seq, _, offs = make_batch(CASES[case], seed=3, letters=SOFT)
    assert run_stats(lib, ctx, seq, offs) == ora.stats(seq, offs)
    assert run_stats(lib, ctx, seq, offs, b"N*") == ora.stats(seq, offs, b"N*")


def test_zero_records_and_bad_arguments(lib, ctx):
    import torch
    offs = to_dev(np.zeros(1, np.uint64))
    buf = torch.zeros(64, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    assert lib.skh_revcomp_dev(ctx, ptr(buf), ptr(buf[16:]), ptr(offs), 0, 0, None, 1, None) == 0
    cls = (C.c_uint8 * 256)(*([0xFF] * 256))
    assert lib.skh_count_dev(ctx, ptr(buf), ptr(offs), 0, 0, cls, 1, None, None) == 0
    assert lib.skh_count_dev(ctx, ptr(buf), ptr(offs), 1, 0, cls, 0, ptr(buf), None) == -1
    assert lib.skh_count_dev(ctx, ptr(buf), ptr(offs), 1, 0, cls, 9, ptr(buf), None) == -1
    assert lib.skh_revcomp_dev(ctx, C.c_void_p(buf.data_ptr() + 1), ptr(buf), ptr(offs), 1, 5, None, 1, None) == -1
    assert b"aligned" in lib.skh_last_error()
    assert lib.skh_revcomp_dev(None, ptr(buf), ptr(buf), ptr(offs), 1, 5, None, 1, None) == -1


# ---------------------------------------------------------------- pinned batch API + fixtures

def fill_batch(lib, b, seq, qual, offs):
    total, nrec = int(offs[-1]), len(offs) - 1
    hs = np.ctypeslib.as_array(C.cast(lib.skh_batch_seq(b), C.POINTER(C.c_uint8)), shape=(lib.skh_batch_max_bytes(b),))
    ho = np.ctypeslib.as_array(C.cast(lib.skh_batch_offsets(b), C.POINTER(C.c_uint64)), shape=(nrec + 1,))
    hs[:total] = seq[:total]
    ho[:] = offs
    hq = None
    if qual is not None:
        hq = np.ctypeslib.as_array(C.cast(lib.skh_batch_qual(b), C.POINTER(C.c_uint8)), shape=(lib.skh_batch_max_bytes(b),))
        hq[:total] = qual[:total]
    return hs, hq


@pytest.mark.parametrize("fixture", ["hairpin.fa.gz", "Illimina1.8.fq.gz", "reads_1.fq.gz", "contigs.fa"])
def test_batch_api_on_reference_fixtures(lib, ctx, ora, fixture):
    names, seq, qual, offs = read_fastx_plain(os.path.join(GOLDEN, fixture))
    total, nrec = int(offs[-1]), len(offs) - 1
    alphabet = 3 if fixture.startswith("hairpin") else 1
    b = C.c_void_p()
    lib.check(lib.skh_batch_alloc(ctx, total, nrec, int(qual is not None), C.byref(b)))
    try:
        hs, hq = fill_batch(lib, b, seq, qual, offs)
        lut = lib.pair_table(alphabet)
        # counts + stats first (they read the batch), then the in-place transform
        cls = np.frombuffer(bytes(lib.class_table(["G", "C", "A", "TU"])), np.uint8)
        counts = np.zeros(nrec * 4, np.uint64)
        lib.check(lib.skh_count(ctx, b, nrec, cls.ctypes.data_as(C.POINTER(C.c_uint8)), 4,
                                counts.ctypes.data_as(C.POINTER(C.c_uint64))))
        assert np.array_equal(counts.reshape(nrec, 4), ora.count(seq, offs, cls, 4))
        from seqkit_amd import StatsPartial
        sp = StatsPartial()
        lib.skh_stats_init(C.byref(sp))
        is_gap = np.zeros(256, np.uint8); is_gap[list(b"- .")] = 1
        lib.check(lib.skh_stats_reduce(ctx, b, nrec, is_gap.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(sp)))
        assert sp.as_tuple() == ora.stats(seq, offs)
        lib.check(lib.skh_revcomp(ctx, b, nrec, lut, 1, 1))
        want_s, want_q = ora.transform(seq, qual, offs, np.frombuffer(bytes(lut), np.uint8), True)
        assert np.array_equal(hs[:total], want_s[:total])
        if qual is not None:
            assert np.array_equal(hq[:total], want_q[:total])
        if fixture.startswith("hairpin"):
            # doc/docs/usage.md:364-366, first record of `seqkit seq hairpin.fa.gz -r -p`
            first = bytes(hs[:int(offs[1])])
            assert first == (b"UCGAAGAGUUCUGUCUCCGGUAAGGUAGAAAAUUGCAUAGUUCACCGGUGGUAAUAUUCC"
                             b"AAACUAUACAACCUACUACCUCACCGGAUCCACAGUGUA")
            assert sp.as_tuple() == (28645, 2949871, 0, 39, 2354)   # doc/docs/usage.md:657
    finally:
        lib.skh_batch_free(b)


def test_batch_capacity_errors(lib, ctx):
    b = C.c_void_p()
    lib.check(lib.skh_batch_alloc(ctx, 64, 4, 0, C.byref(b)))
    try:
        ho = np.ctypeslib.as_array(C.cast(lib.skh_batch_offsets(b), C.POINTER(C.c_uint64)), shape=(5,))
        ho[:] = [0, 10, 20, 30, 100]
        assert lib.skh_revcomp(ctx, b, 4, None, 1, 0) == -5
        assert lib.skh_revcomp(ctx, b, 5, None, 1, 0) == -5
    finally:
        lib.skh_batch_free(b)


def test_two_batches_in_flight(lib, ctx, ora):
    # split-phase submit/wait on two batches, as a double-buffering reader would use them
    batches, data = [], []
    for k in range(2):
        seq, qual, offs = make_batch(np.full(3000, 150) if k == 0 else CASES["ragged0_300"], seed=50 + k,
                                     letters=b"ACGTN", with_qual=True)
        b = C.c_void_p()
        lib.check(lib.skh_batch_alloc(ctx, int(offs[-1]), len(offs) - 1, 1, C.byref(b)))
        hs, hq = fill_batch(lib, b, seq, qual, offs)
        batches.append(b); data.append((seq, qual, offs, hs, hq))
    lut = lib.pair_table(1)
    try:
        for b, d in zip(batches, data):
            lib.check(lib.skh_revcomp_submit(ctx, b, len(d[2]) - 1, lut, 1, 1))
        for b, d in zip(batches, data):
            lib.check(lib.skh_batch_wait(b))
            seq, qual, offs, hs, hq = d
            total = int(offs[-1])
            want_s, want_q = ora.transform(seq, qual, offs, np.frombuffer(bytes(lut), np.uint8), True)
            assert np.array_equal(hs[:total], want_s[:total]) and np.array_equal(hq[:total], want_q[:total])
    finally:
        for b in batches:
            lib.skh_batch_free(b)


def test_round_trip_at_bench_size(lib, ctx):
    # size-independent property at a large size: revcomp twice is the identity, and the
    # counts of the reverse complement are the complement-swapped counts of the input
    import torch
    nrec, L = 2_000_000, 150
    g = torch.Generator(device="cuda:0"); g.manual_seed(11)
    letters = torch.tensor(list(b"ACGTN"), dtype=torch.uint8, device="cuda:0")
    seq = letters[torch.randint(0, 5, (nrec * L,), generator=g, device="cuda:0")]
    offs = (torch.arange(nrec + 1, dtype=torch.int64, device="cuda:0") * L)
    a, b2 = torch.empty_like(seq), torch.empty_like(seq)
    lut = lib.pair_table(1)
    torch.cuda.synchronize()
    lib.check(lib.skh_revcomp_dev(ctx, ptr(seq), ptr(a), ptr(offs), nrec, nrec * L, lut, 1, None))
    lib.check(lib.skh_revcomp_dev(ctx, ptr(a), ptr(b2), ptr(offs), nrec, nrec * L, lut, 1, None))
    cls = lib.class_table(["A", "C", "G", "T"])
    c1 = torch.empty(nrec * 4, dtype=torch.int64, device="cuda:0"); c2 = torch.empty_like(c1)
    lib.check(lib.skh_count_dev(ctx, ptr(seq), ptr(offs), nrec, nrec * L, cls, 4, ptr(c1), None))
    lib.check(lib.skh_count_dev(ctx, ptr(a), ptr(offs), nrec, nrec * L, cls, 4, ptr(c2), None))
    lib.check(lib.skh_sync(ctx))
    assert torch.equal(seq, b2)
    assert not torch.equal(seq, a)
    c1, c2 = c1.view(nrec, 4), c2.view(nrec, 4)
    assert torch.equal(c1[:, [3, 2, 1, 0]], c2)           # A<->T, C<->G
    assert int(c1.sum()) == int((seq != ord("N")).sum())


@pytest.mark.parametrize("shape", ["reads100", "trimmed30_151", "heavy_tail", "kilobases", "hairpin_like", "reads_and_contigs",
                                   "reads_and_12kb"])
def test_round_trip_at_scale_ragged(lib, ctx_default, shape):
    # Size-independent properties on batches of 0.2-0.5 Gbases whose lengths are not the bench's 150: reversing and
    # complementing twice is the identity (out of place and in place), the first pass equals the in-place pass, and
    # the per-record counts of the result are the complement-swapped counts of the input.  First and last records
    # against numpy.
    import torch
    rng = np.random.default_rng(17)
    lengths = {"reads100": np.full(3_000_000, 100), "trimmed30_151": rng.integers(30, 152, 3_000_000),
               "heavy_tail": np.clip(np.exp(rng.normal(np.log(120), 0.7, 2_000_000)), 1, 40000).astype(np.int64),
               "kilobases": rng.integers(500, 2500, 300_000),
               # precursor-like lengths around 100 with a tail to 2354 (the flat kernel's overflow steps by lane groups)
               "hairpin_like": np.clip(np.exp(rng.normal(np.log(100), 0.35, 3_000_000)), 39, 2354),
               # reads with a few long contigs among them (the launch's list of long records), and with one record in
               # ten of 12 kb (steps that overflow the stage served by whole wavefronts)
               "reads_and_contigs": np.where(rng.random(1_500_000) < 1e-4, rng.integers(100_000, 3_000_000, 1_500_000), 150),
               "reads_and_12kb": np.where(rng.random(600_000) < 0.1, 12000, 150)}[shape].astype(np.int64)
    offs_h = np.zeros(len(lengths) + 1, np.int64); np.cumsum(lengths, out=offs_h[1:])
    nrec, total = len(lengths), int(offs_h[-1])
    g = torch.Generator(device="cuda:0"); g.manual_seed(3)
    letters = torch.tensor(list(b"ACGTNacgtn"), dtype=torch.uint8, device="cuda:0")
    seq = torch.zeros((total + 15) & ~15, dtype=torch.uint8, device="cuda:0")
    seq[:total] = letters[torch.randint(0, len(letters), (total,), generator=g, device="cuda:0")]
    offs = torch.from_numpy(offs_h).to("cuda:0")
    a, b2, w = torch.empty_like(seq), torch.empty_like(seq), seq.clone()
    lut = lib.pair_table(1)
    torch.cuda.synchronize()
    lib.check(lib.skh_revcomp_dev(ctx_default, ptr(seq), ptr(a), ptr(offs), nrec, total, lut, 1, None))
    lib.check(lib.skh_revcomp_dev(ctx_default, ptr(a), ptr(b2), ptr(offs), nrec, total, lut, 1, None))
    lib.check(lib.skh_revcomp_inplace_dev(ctx_default, ptr(w), None, ptr(offs), nrec, total, lut, 1, None))
    cls = lib.class_table(["Aa", "Cc", "Gg", "Tt"])
    c1 = torch.empty(nrec * 4, dtype=torch.int64, device="cuda:0"); c2 = torch.empty_like(c1)
    lib.check(lib.skh_count_dev(ctx_default, ptr(seq), ptr(offs), nrec, total, cls, 4, ptr(c1), None))
    lib.check(lib.skh_count_dev(ctx_default, ptr(a), ptr(offs), nrec, total, cls, 4, ptr(c2), None))
    lib.check(lib.skh_sync(ctx_default))
    assert torch.equal(seq[:total], b2[:total])
    assert torch.equal(a[:total], w[:total])
    lib.check(lib.skh_revcomp_inplace_dev(ctx_default, ptr(w), None, ptr(offs), nrec, total, lut, 1, None))
    lib.check(lib.skh_sync(ctx_default))
    assert torch.equal(seq[:total], w[:total])
    c1, c2 = c1.view(nrec, 4), c2.view(nrec, 4)
    assert torch.equal(c1[:, [3, 2, 1, 0]], c2)           # A<->T, C<->G
    lut_np = np.frombuffer(bytes(lut), np.uint8)
    for r in (0, 1, nrec // 2, nrec - 2, nrec - 1):
        s, e = int(offs_h[r]), int(offs_h[r + 1])
        assert np.array_equal(a[s:e].cpu().numpy(), lut_np[seq[s:e].cpu().numpy()[::-1]]), r
