"""Several GPUs as the library and the driver see them.  The test box has one GPU, so the
many-context paths run with contexts that share it (`--devices 0,0`: blocks dealt to two contexts,
their partial totals merged, the RCCL communicator made over the one distinct device) and the
multi-process form runs as a job of one rank; with two or more GPUs visible the same tests also run
over distinct devices.  Every result must equal the single-GPU result and the oracle's."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT, run_cli
from helpers import Oracle, make_batch, ptr, to_dev, wrapped_fastq

pytestmark = pytest.mark.gpu

CLI = os.path.join(ROOT, "seqkit_amd", "seqkit-hip")


def device_sets():
    import torch
    sets = [["--devices", "0,0"], ["--devices", "0,0,0"]]
    if torch.cuda.device_count() >= 2:
        sets += [["--gpus", "2"], ["--devices", "1,0"]]
    return sets


COMMANDS = [(["seq", "-r", "-p"], ["hairpin.fa.gz"]), (["seq", "-r", "-p", "-w", "0"], ["reads_1.fq.gz"]),
            (["seq"], ["reads_1.fq.gz", "contigs.fa"]), (["fx2tab", "-n", "-l", "-g", "-B", "AGCT"], ["Illimina1.8.fq.gz"]),
            (["stats"], ["hairpin.fa.gz", "mature.fa.gz", "Illimina1.8.fq.gz", "reads_1.fq.gz", "reads_2.fq.gz"]),
            (["stats", "-a", "-T"], ["hairpin.fa.gz", "contigs.fa"]), (["subseq", "-r", "2:-2"], ["mature.fa.gz"]),
            (["convert", "--to", "Illumina-1.5+"], ["reads_2.fq.gz"])]


@pytest.mark.parametrize("argv,files", COMMANDS, ids=lambda v: " ".join(v) if isinstance(v, list) and v and v[0][0] != "-" else None)
def test_driver_deals_blocks_to_several_contexts(oracle_cli, argv, files):
    want = run_cli(oracle_cli, argv + files)
    single = run_cli(CLI, argv + ["--batch-bytes", "30000"] + files)
    assert single[0] == want[0] == 0 and single[1] == want[1]
    for dev in device_sets():
        got = run_cli(CLI, argv + dev + ["--batch-bytes", "30000"] + files)
        assert got[0] == 0 and got[1] == want[1], (dev, got[2][-400:])


def test_driver_multiline_fastq_over_several_contexts(oracle_cli, tmp_path):
    f = str(tmp_path / "late.fq")
    wrapped_fastq(f, four_line_head=1200)
    for argv in (["seq", "-r", "-p"], ["stats", "-a"]):
        want = run_cli(oracle_cli, argv + [f])
        got = run_cli(CLI, argv + ["--devices", "0,0", "--batch-bytes", "25000", f])
        assert got[0] == 0 and got[1] == want[1]


def test_one_record_cut_over_three_contexts(oracle_cli, tmp_path):
    """`--devices 0,0,0` on a file that is one 5 Mb record, and on a file with long and short records mixed:
    the file goes through a pinned batch that is cut over all contexts (inside the long records), and the
    output is the oracle's byte for byte.  SKH_SHARES=1 makes the driver print what every context was given."""
    rng = np.random.default_rng(4)
    files = {}
    for name, lens in (("one.fa", [5_000_003]), ("mixed_long.fa", [1_200_000, 12, 0, 700_001, 90_000, 5] + [150] * 3000)):
        path = str(tmp_path / name)
        with open(path, "wb") as fh:
            for i, n in enumerate(lens):
                s = rng.choice(np.frombuffer(b"ACGTacgtNn-", np.uint8), size=n).tobytes()
                fh.write(b">c%d some text\n" % i)
                for a in range(0, n, 60):
                    fh.write(s[a:a + 60] + b"\n")
        files[name] = path
    env = dict(os.environ, SKH_SHARES="1")
    import subprocess
    for argv in (["seq", "-r", "-p"], ["seq", "-r", "-p", "-w", "0"], ["fx2tab", "-n", "-l", "-g", "-B", "AGCT"], ["stats", "-a", "-T"]):
        for name, path in files.items():
            want = run_cli(oracle_cli, argv + [path])
            for extra in (["--batch-bytes", "1000000"], ["--host-parse"]):
                p = subprocess.run([CLI] + argv + ["--devices", "0,0,0"] + extra + [path], stdout=subprocess.PIPE,
                                   stderr=subprocess.PIPE, env=env, timeout=300)
                assert p.returncode == 0 and p.stdout == want[1], (argv, name, extra, p.stderr[-400:])
                shares = [l for l in p.stderr.decode().splitlines() if l.startswith("[shares]")]
                assert len(shares) >= 3, p.stderr[-400:]
                if name == "one.fa":
                    # all three contexts worked on the one record: two of the shares start inside it
                    first = shares[:3]
                    assert all("records 0..1" in l for l in first) and sum("head_cut 0 " not in l for l in first) == 2
    # one context: the same files stay on the text path (nothing is cut)
    p = subprocess.run([CLI, "seq", "-r", "-p", "--batch-bytes", "1000000", files["one.fa"]], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, env=env, timeout=300)
    assert p.returncode == 0 and b"[shares]" not in p.stderr
    assert p.stdout == run_cli(oracle_cli, ["seq", "-r", "-p", files["one.fa"]])[1]


def test_fasta_after_fastq_and_stats_type(oracle_cli):
    want = run_cli(oracle_cli, ["seq", "reads_1.fq.gz", "contigs.fa"])
    assert want[1].endswith(b"@length=9\nAAAAAAAAA\n+\n\n")
    for mode in ([], ["--host-parse"]):
        got = run_cli(CLI, ["seq"] + mode + ["reads_1.fq.gz", "contigs.fa"])
        assert got[0] == 0 and got[1] == want[1]
        got = run_cli(CLI, ["stats", "-T", "-t", "protein"] + mode + ["contigs.fa"])
        assert got[1] == run_cli(oracle_cli, ["stats", "-T", "-t", "protein", "contigs.fa"])[1]


def test_stats_allreduce_host_and_device_forms(oracle_lib):
    import torch
    from seqkit_amd import StatsPartial, load
    lib = load()
    ora = Oracle(oracle_lib)
    ndev = torch.cuda.device_count()
    devs = [0, 0, 0] if ndev < 2 else [0, 1, 0]
    ctxs = [lib.init(d) for d in devs]
    gap = (C.c_uint8 * 256)()
    for ch in b"- .":
        gap[ch] = 1
    try:
        # three shares of one batch, one per context: partial totals, then the collective
        lengths = np.concatenate([np.random.default_rng(3).integers(0, 500, 3000), [70000, 0, 12]])
        seq, _, offs = make_batch(lengths, seed=8, letters=b"ACGT-. N")
        nrec = len(lengths)
        cuts = np.zeros(4, np.uint64)
        lib.check(lib.skh_split_by_bytes(offs.ctypes.data_as(C.POINTER(C.c_uint64)), nrec, 3, cuts.ctypes.data_as(C.POINTER(C.c_uint64))))
        parts = (StatsPartial * 3)()
        dev_parts = []
        for k in range(3):
            lo, hi = int(cuts[k]), int(cuts[k + 1])
            sub = np.ascontiguousarray(seq[int(offs[lo]):int(offs[hi])])
            pad = np.zeros(max((len(sub) + 15) & ~15, 16), np.uint8); pad[:len(sub)] = sub
            so = (offs[lo:hi + 1] - offs[lo]).astype(np.uint64)
            dname = "cuda:%d" % devs[k]
            d_seq, d_offs = to_dev(pad, dname), to_dev(so, dname)
            d_out = torch.zeros(5, dtype=torch.int64, device=dname)
            torch.cuda.synchronize(dname)
            lib.check(lib.skh_stats_dev(ctxs[k], ptr(d_seq), ptr(d_offs), hi - lo, int(so[-1]), gap, ptr(d_out), None))
            lib.check(lib.skh_sync(ctxs[k]))
            v = [int(x) for x in d_out.cpu()]
            parts[k].num, parts[k].len_sum, parts[k].gap_sum, parts[k].len_min, parts[k].len_max = v
            dev_parts.append(d_out)
        comm = C.c_void_p()
        arr = (C.c_void_p * 3)(*[c.value for c in ctxs])
        lib.check(lib.skh_comm_init_all(arr, 3, C.byref(comm)))
        assert lib.skh_comm_local_ranks(comm) == len(set(devs))
        lib.check(lib.skh_stats_allreduce(comm, parts))
        want = ora.stats(seq, offs)
        for k in range(3):
            assert parts[k].as_tuple() == want
        lib.skh_comm_destroy(comm)
        # the multi-process form as a job of one rank, device buffers reduced in place
        raw = (C.c_uint8 * 128)()
        lib.check(lib.skh_comm_unique_id(raw))
        lib.check(lib.skh_comm_init_rank(ctxs[0], raw, 1, 0, C.byref(comm)))
        before = [int(x) for x in dev_parts[0].cpu()]
        dp = (C.c_void_p * 1)(dev_parts[0].data_ptr())
        lib.check(lib.skh_stats_allreduce_dev(comm, dp, None))
        torch.cuda.synchronize()
        assert [int(x) for x in dev_parts[0].cpu()] == before
        lib.skh_comm_destroy(comm)
    finally:
        for c in ctxs:
            lib.skh_destroy(c)
