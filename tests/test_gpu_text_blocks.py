"""The text-block entry points (skh_text_*) called through the C ABI the way a cgo shim would:
file text in a pinned block, parse on the GPU, hot-path kernels, output text back.  Results are
compared with the oracle's command-line restatement on the same text and with the oracle's batch
functions on the parsed planes.  skh_text_parse_at is run with the text at several offsets of
the block: the outcome must not depend on where the text starts."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import run_cli
from helpers import Oracle, read_fastx_plain

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from seqkit_amd import load
    return load()


@pytest.fixture(scope="module")
def ctx(lib):
    c = lib.init(0)
    yield c
    lib.skh_destroy(c)


@pytest.fixture(scope="module")
def ora(oracle_lib):
    return Oracle(oracle_lib)


class Block:
    def __init__(self, lib, ctx, nbytes):
        self.lib, self.ctx = lib, ctx
        self.t = C.c_void_p()
        lib.check(lib.skh_text_alloc(ctx, nbytes, C.byref(self.t)))
        self.cap = lib.skh_text_capacity(self.t)
        self.buf = np.ctypeslib.as_array(C.cast(lib.skh_text_buffer(self.t), C.POINTER(C.c_uint8)), shape=(self.cap,))

    def close(self):
        self.lib.skh_text_free(self.t)

    def parse_at(self, text, off, fmt, final=1, ahead=None):
        from seqkit_amd.capi import ParseResult
        self.buf[:] = 0x40                          # '@': stale bytes around the text must not matter
        self.buf[off:off + len(text)] = np.frombuffer(text, np.uint8)
        if ahead is not None:                       # (offset, nbytes) sent ahead of the parse, as a reader does
            self.lib.check(self.lib.skh_text_upload(self.ctx, self.t, ahead[0], ahead[1]))
        res = ParseResult()
        self.lib.check(self.lib.skh_text_parse_at(self.ctx, self.t, off, len(text), fmt, final, C.byref(res)))
        return res

    def names(self, n):
        pos = np.zeros(n, np.uint64)
        ln = np.zeros(n, np.uint32)
        self.lib.check(self.lib.skh_text_names(self.ctx, self.t, 0, n, pos.ctypes.data_as(C.POINTER(C.c_uint64)),
                                               ln.ctypes.data_as(C.POINTER(C.c_uint32))))
        return [bytes(self.buf[int(p):int(p) + int(l)]) for p, l in zip(pos, ln)]

    def format_seq(self, **kw):
        from seqkit_amd.capi import SeqFormat
        f = SeqFormat(min_len=-1, max_len=-1, line_width=kw.get("width", 0), print_name=1, print_seq=1,
                      print_qual=kw.get("qual", 0), plus_line=1, only_id=0, mark=kw.get("mark", b">"), empty_as_fasta=0)
        out, n = C.c_void_p(), C.c_uint64()
        self.lib.check(self.lib.skh_text_format_seq(self.ctx, self.t, C.byref(f), C.byref(out), C.byref(n)))
        return C.string_at(out, n.value)


def fixture_text(plain_fixtures, name):
    with open(plain_fixtures[name], "rb") as fh:
        return fh.read()


@pytest.mark.parametrize("name,fmt", [("hairpin.fa", 1), ("Illimina1.8.fq", 2), ("contigs.fa", 1)])
def test_text_block_matches_oracle_at_any_offset(lib, ctx, ora, oracle_cli, plain_fixtures, name, fmt):
    from seqkit_amd.capi import StatsPartial
    text = fixture_text(plain_fixtures, name)
    names, seq, qual, offs = read_fastx_plain(plain_fixtures[name])
    nrec, total = len(offs) - 1, int(offs[-1])
    rc, want_rc_text, _ = run_cli(oracle_cli, ["seq", "-r", "-p", plain_fixtures[name]])
    assert rc == 0
    alphabet = 3 if name.startswith("hairpin") else 1
    lut = lib.pair_table(alphabet)
    blk = Block(lib, ctx, len(text) + (1 << 16))
    try:
        for off in (0, 1, 15, 16, 4097, 65519):
            res = blk.parse_at(text, off, fmt)
            assert (res.status, res.nrec, res.seq_bytes, res.consumed) == (0, nrec, total, len(text))
            assert blk.names(nrec) == names
            got_offs = np.zeros(nrec + 1, np.uint64)
            lib.check(lib.skh_text_offsets(ctx, blk.t, 0, nrec + 1, got_offs.ctypes.data_as(C.POINTER(C.c_uint64))))
            assert np.array_equal(got_offs, offs)
            got_seq = np.zeros(total, np.uint8)
            lib.check(lib.skh_text_sequence(ctx, blk.t, 0, total, got_seq.ctypes.data_as(C.POINTER(C.c_uint8))))
            assert np.array_equal(got_seq, seq[:total])
            sp = StatsPartial()
            lib.skh_stats_init(C.byref(sp))
            is_gap = (C.c_uint8 * 256)()
            for ch in b"- .":
                is_gap[ch] = 1
            lib.check(lib.skh_text_stats(ctx, blk.t, is_gap, C.byref(sp)))
            assert sp.as_tuple() == ora.stats(seq, offs)
            # seq -r -p: reverse + complement on the block, output text from the GPU formatter
            lib.check(lib.skh_text_revcomp(ctx, blk.t, lut, 1, 1))
            got = blk.format_seq(width=0 if fmt == 2 else 60, qual=int(fmt == 2), mark=b"@" if fmt == 2 else b">")
            assert got == want_rc_text
    finally:
        blk.close()


def test_blockwise_tail_carry(lib, ctx, plain_fixtures):
    # a file fed in pieces the way a run-ahead reader does it: every piece is read into the block
    # some headroom in, the unconsumed tail of the piece before is put in front of it
    text = b"\n".join(fixture_text(plain_fixtures, "Illimina1.8.fq").split(b"\n")[:4 * 1200]) + b"\n"
    _, _, _, offs = read_fastx_plain_bytes(text)
    piece, head = 30_011, 4096
    blk = Block(lib, ctx, piece + head + 64)
    lens, tail, pos = [], b"", 0
    try:
        while pos < len(text) or tail:
            body = text[pos:pos + piece]
            pos += len(body)
            final = int(pos >= len(text))
            assert len(tail) <= head
            chunk = tail + body
            # every other piece: the body goes ahead (skh_text_upload), the tail follows with the parse;
            # now and then a range that the text does not cover, which the parse has to ignore
            step = len(lens) % 3
            ahead = None if step == 0 else (head, len(body)) if step == 1 else (0, 64)
            res = blk.parse_at(chunk, head - len(tail), 2, final, ahead)
            assert res.status == 0
            o = np.zeros(res.nrec + 1, np.uint64)
            lib.check(lib.skh_text_offsets(ctx, blk.t, 0, res.nrec + 1, o.ctypes.data_as(C.POINTER(C.c_uint64))))
            lens += list(np.diff(o))
            tail = b"" if final else chunk[res.consumed:]
            if final:
                break
    finally:
        blk.close()
    assert lens == list(np.diff(offs))


def read_fastx_plain_bytes(data):
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".fq") as fh:
        fh.write(data)
        fh.flush()
        return read_fastx_plain(fh.name)


def test_parse_at_rejects_text_beyond_the_buffer(lib, ctx):
    from seqkit_amd.capi import ParseResult
    t = C.c_void_p()
    lib.check(lib.skh_text_alloc(ctx, 8192, C.byref(t)))
    try:
        cap = lib.skh_text_capacity(t)
        res = ParseResult()
        assert lib.skh_text_parse_at(ctx, t, cap - 10, 11, 1, 1, C.byref(res)) == -5
        assert lib.skh_text_parse_at(ctx, t, cap + 1, 0, 1, 1, C.byref(res)) == -5
    finally:
        lib.skh_text_free(t)


def test_format_submit_then_wait_equals_the_blocking_call(lib, ctx, plain_fixtures):
    # two blocks: the first one's output is on its way while the second one is parsed and formatted
    from seqkit_amd.capi import SeqFormat
    text_a = fixture_text(plain_fixtures, "Illimina1.8.fq")
    text_b = fixture_text(plain_fixtures, "reads_1.fq")
    a, b = Block(lib, ctx, len(text_a) + 64), Block(lib, ctx, len(text_b) + 64)
    try:
        f = SeqFormat(min_len=-1, max_len=-1, line_width=0, print_name=1, print_seq=1, print_qual=1, plus_line=1,
                      only_id=0, mark=b"@", empty_as_fasta=0)
        assert a.parse_at(text_a, 0, 2).status == 0
        want_a = a.format_seq(qual=1, mark=b"@")
        lib.check(lib.skh_text_format_seq_submit(ctx, a.t, C.byref(f)))
        assert b.parse_at(text_b, 0, 2).status == 0
        want_b = b.format_seq(qual=1, mark=b"@")
        lib.check(lib.skh_text_format_seq_submit(ctx, b.t, C.byref(f)))
        out, n = C.c_void_p(), C.c_uint64()
        lib.check(lib.skh_text_output_wait(ctx, a.t, C.byref(out), C.byref(n)))
        assert C.string_at(out, n.value) == want_a
        lib.check(lib.skh_text_output_wait(ctx, b.t, C.byref(out), C.byref(n)))
        assert C.string_at(out, n.value) == want_b
        lib.check(lib.skh_text_output_wait(ctx, b.t, C.byref(out), C.byref(n)))      # nothing pending any more
        assert n.value == 0
    finally:
        a.close()
        b.close()
