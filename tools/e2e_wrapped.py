#!/usr/bin/env python3
"""End to end on wrapped FASTQ: NREC records of 150 bases wrapped at 60 columns through `seqkit-hip seq -r -p`
with the GPU's parser of wrapped FASTQ (default) and with the serial rewrite alone (SKH_FASTQ_REWRITE=host),
against the oracle CLI (md5 of the outputs).  Page cache warm; timed with the output to /dev/null (process start to exit), md5 from a run of its own."""
import hashlib, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
nrec = int(os.environ.get("NREC", 2_000_000)); L, W = 150, 60
rng = np.random.default_rng(2)
head = b"@read 1:N:0:ACGT\n"
nl = (L + W - 1) // W
rec_len = len(head) + 2 * (L + nl) + 2
rec = np.zeros(rec_len, np.uint8); rec[:len(head)] = np.frombuffer(head, np.uint8)
text = np.tile(rec, nrec).reshape(nrec, rec_len)
seq = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (nrec, L))]
qual = rng.integers(33, 74, (nrec, L), dtype=np.uint8)
p = len(head)
for a in range(0, L, W):
    k = min(W, L - a); text[:, p:p + k] = seq[:, a:a + k]; text[:, p + k] = 10; p += k + 1
text[:, p] = ord("+"); text[:, p + 1] = 10; p += 2
for a in range(0, L, W):
    k = min(W, L - a); text[:, p:p + k] = qual[:, a:a + k]; text[:, p + k] = 10; p += k + 1
path = "/tmp/wrapped_e2e.fq"
text.reshape(-1).tofile(path)
size = os.path.getsize(path)


def run(cmd, env=None, keep=False):
    """keep: the output goes through a pipe into md5 (not timed); else to /dev/null, process start to exit"""
    e = dict(os.environ); e.update(env or {})
    t0 = time.perf_counter()
    out = subprocess.run(cmd, stdout=subprocess.PIPE if keep else subprocess.DEVNULL, stderr=subprocess.PIPE, env=e)
    dt = time.perf_counter() - t0
    return out.returncode, hashlib.md5(out.stdout).hexdigest() if keep else None, dt, out.stderr[-300:]


drv = os.path.join(ROOT, "seqkit_amd", "seqkit-hip"); ora = os.path.join(ROOT, "oracle", "seqkit_oracle")
print("%d records x %d bases wrapped at %d columns: %.2f GB of text" % (nrec, L, W, size / 1e9))
res = {}
for name, cmd, env in (("seqkit-hip, GPU parser of wrapped FASTQ", [drv, "seq", "-r", "-p", path], {}),
                       ("seqkit-hip, SKH_FASTQ_REWRITE=host", [drv, "seq", "-r", "-p", path], {"SKH_FASTQ_REWRITE": "host"}),
                       ("oracle CLI (plain C, 1 thread)", [ora, "seq", "-r", "-p", path], {})):
    rc, md5, _, err = run(cmd, env, keep=True)
    assert rc == 0, err
    best = None
    for _ in range(2 if "oracle" in name else 3):
        rc, _, dt, err = run(cmd, env)
        assert rc == 0, err
        best = dt if best is None or dt < best else best
    res[name] = md5
    print("   %-44s %7.3f s  %6.2f Gbases/s  md5 %s" % (name, best, nrec * L / best / 1e9, md5))
if not os.environ.get("KEEP"): os.remove(path)
ok = len(set(res.values())) == 1
print("   outputs identical:", ok)
sys.exit(0 if ok else 1)
