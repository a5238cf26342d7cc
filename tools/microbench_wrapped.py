#!/usr/bin/env python3
"""skh_parse_dev with SKH_FASTQ_WRAPPED on FASTQ whose sequence and quality are wrapped at 60 columns:
NREC records of READLEN bases (default 1 M x 150), resident in HBM; for comparison the four-line parser
on the four-line form of the same records."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import seqkit_amd
from seqkit_amd import capi

lib = seqkit_amd.load(); ctx = lib.init(0); dev = torch.device("cuda", 0)
nrec = int(os.environ.get("NREC", 1_000_000)); L = int(os.environ.get("READLEN", 150)); W = 60
rng = np.random.default_rng(1)


def make(width):
    head = b"@read 1:N:0:ACGT\n"
    nl = (L + width - 1) // width
    rec_len = len(head) + 2 * (L + nl) + 2
    rec = np.zeros(rec_len, np.uint8)
    rec[:len(head)] = np.frombuffer(head, np.uint8)
    text = np.tile(rec, nrec).reshape(nrec, rec_len)
    seq = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (nrec, L))]
    qual = rng.integers(33, 74, (nrec, L), dtype=np.uint8)
    p = len(head)
    for a in range(0, L, width):
        k = min(width, L - a); text[:, p:p + k] = seq[:, a:a + k]; text[:, p + k] = 10; p += k + 1
    text[:, p] = ord("+"); text[:, p + 1] = 10; p += 2
    for a in range(0, L, width):
        k = min(width, L - a); text[:, p:p + k] = qual[:, a:a + k]; text[:, p + k] = 10; p += k + 1
    assert p == rec_len
    return text.reshape(-1), seq.reshape(-1), qual.reshape(-1)


def run(name, fmt, text, seq, qual):
    n = len(text); lines = int((text == 10).sum()) + 2
    pad = (n + 15) & ~15
    d_text = torch.zeros(pad, dtype=torch.uint8, device=dev); d_text[:n] = torch.from_numpy(text).to(dev)
    d_seq = torch.zeros(pad, dtype=torch.uint8, device=dev); d_qual = torch.zeros(pad, dtype=torch.uint8, device=dev)
    d_offs = torch.zeros(nrec + 2, dtype=torch.int64, device=dev)
    d_np = torch.zeros(nrec + 1, dtype=torch.int64, device=dev); d_nl = torch.zeros(nrec + 1, dtype=torch.int32, device=dev)
    wsb = lib.skh_parse_wrapped_workspace_size(n, lines, nrec + 1) if fmt == capi.FASTQ_WRAPPED else lib.skh_parse_workspace_size(n, lines)
    d_ws = torch.zeros(wsb + 256, dtype=torch.uint8, device=dev); ws = (d_ws.data_ptr() + 255) & ~255
    d_res = torch.zeros(6, dtype=torch.int64, device=dev)
    st = torch.cuda.Stream(device=dev); sp = C.c_void_p(st.cuda_stream)
    torch.cuda.synchronize()

    def fn():
        lib.check(lib.skh_parse_dev(ctx, d_text.data_ptr(), n, fmt, 1, d_seq.data_ptr(), d_qual.data_ptr(), pad, d_offs.data_ptr(),
                                    d_np.data_ptr(), d_nl.data_ptr(), nrec + 1, ws, wsb, lines, d_res.data_ptr(), sp))
    with torch.cuda.stream(st):
        for _ in range(2): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(5): fn()
        e1.record(st); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    res = capi.ParseResult.from_buffer_copy(d_res.cpu().numpy().tobytes())
    ok = res.status == 0 and res.nrec == nrec and bool(torch.equal(d_seq[:nrec * L].cpu(), torch.from_numpy(seq))) and \
        bool(torch.equal(d_qual[:nrec * L].cpu(), torch.from_numpy(qual)))
    print("%-34s %8.3f ms  %7.1f GB/s of text  workspace %.2f GB  %s" % (name, ms, n / ms / 1e6, wsb / 1e9, "ok" if ok else "MISMATCH"))
    return ok


t, s, q = make(W)
ok = run("wrapped at 60, SKH_FASTQ_WRAPPED", capi.FASTQ_WRAPPED, t, s, q)
t4, s4, q4 = make(1 << 30)
ok &= run("four-line, SKH_FASTQ_WRAPPED", capi.FASTQ_WRAPPED, t4, s4, q4)
ok &= run("four-line, SKH_FASTQ", capi.FASTQ, t4, s4, q4)
lib.skh_destroy(ctx)
sys.exit(0 if ok else 1)
