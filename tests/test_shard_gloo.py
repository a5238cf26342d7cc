"""The N>1 path on CPU: byte-volume sharding, intra-record slicing and the stats all-reduce,
two ranks over gloo, with the oracle standing in for each rank's GPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, build_oracle
from helpers import Oracle, make_batch
from seqkit_amd import shard


def test_split_by_bytes_is_contiguous_and_balanced():
    rng = np.random.default_rng(0)
    lengths = np.concatenate([rng.integers(0, 400, 5000), [50000, 0, 0, 7]])
    offs = np.zeros(len(lengths) + 1, np.uint64)
    np.cumsum(lengths, out=offs[1:])
    for n in (1, 2, 3, 4, 8):
        parts = shard.split_by_bytes(offs, n)
        assert parts[0][0] == 0 and parts[-1][1] == len(lengths)
        assert all(parts[i][1] == parts[i + 1][0] for i in range(n - 1))
        vol = [int(offs[b] - offs[a]) for a, b in parts]
        assert sum(vol) == int(offs[-1])
        assert max(vol) - min(vol) <= 2 * 50000
    # more ranks than records, and an empty batch
    assert shard.split_by_bytes(np.array([0, 5], np.uint64), 4)[-1][1] == 1
    assert shard.split_by_bytes(np.array([0], np.uint64), 2) == [(0, 0), (0, 0)]


def test_split_record_reassembles_to_revcomp():
    ora = Oracle(__import__("ctypes").CDLL(build_oracle()[1]))
    seq, _, offs = make_batch([100003], seed=4, letters=b"ACGTNacgtn")
    lut = ora.pair_table(1)
    want, _ = ora.transform(seq, None, offs, lut, True)
    L = 100003
    for n in (2, 4, 8):
        out = np.zeros(L, np.uint8)
        for a, b, pos in shard.split_record(L, n):
            assert a % 16 == 0 or a == 0
            piece = np.zeros(max((b - a + 15) & ~15, 16), np.uint8)
            piece[:b - a] = seq[a:b]
            o = np.array([0, b - a], np.uint64)
            got, _ = ora.transform(piece, None, o, lut, True)   # a slice is just a record
            out[pos:pos + (b - a)] = got[:b - a]
        assert np.array_equal(out, want[:L])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, lengths, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ctypes
    ora = Oracle(ctypes.CDLL(os.path.join(ROOT, "oracle", "libseqkit_oracle.so")))
    seq, _, offs = make_batch(lengths, seed=9, letters=b"ACGT-.N ")
    lo, hi = shard.split_by_bytes(offs, world)[rank]
    sub_offs = (offs[lo:hi + 1] - offs[lo]).astype(np.uint64)
    sub_seq = np.ascontiguousarray(seq[int(offs[lo]):int(offs[hi])])
    part = ora.stats(sub_seq, sub_offs) if hi > lo else (0, 0, 0, 2**63 - 1, 0)
    tot = shard.allreduce_stats(torch.tensor(part, dtype=torch.int64))
    q.put((rank, [int(x) for x in tot], list(ora.stats(seq, offs))))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("lengths", [list(range(0, 300, 7)) * 20, [5], [0, 0, 9, 100000, 3]])
def test_stats_allreduce_two_ranks(lengths):
    build_oracle()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, lengths, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, tot, want in res:
        assert tot == want, (rank, tot, want)
