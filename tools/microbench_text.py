#!/usr/bin/env python3
"""Times the text stages on a resident block (GPU only): FASTA/FASTQ parse, seq output
formatting and fx2tab row formatting, with their algorithmic bytes (text read once, planes
written once, and the reverse for formatting)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import seqkit_amd
from seqkit_amd import capi

nrec = int(os.environ.get("NREC", 4_000_000)); L = int(os.environ.get("READLEN", 150))
dev = torch.device("cuda", 0)
lib = seqkit_amd.load(); ctx = lib.init(0)


def aligned(nbytes):
    t = torch.zeros(nbytes + 256, dtype=torch.uint8, device=dev)
    return t, (t.data_ptr() + 255) & ~255


def run(kind):
    path = "/tmp/mbt.%s" % kind
    args = [sys.executable, os.path.join(ROOT, "tools", "gen_fastx.py"), path, str(nrec), str(L)]
    if kind == "fa":
        args += ["0xC3", "--fasta", "--width", "60"]
    subprocess.check_call(args)
    text = np.fromfile(path, np.uint8)
    n = len(text)
    padded = (n + 15) & ~15
    h = np.zeros(padded, np.uint8); h[:n] = text
    d_text = torch.from_numpy(h).to(dev)
    fastq = kind == "fq"
    max_lines = n // 32 + 1024
    d_seq = torch.zeros(padded, dtype=torch.uint8, device=dev)
    d_qual = torch.zeros(padded, dtype=torch.uint8, device=dev)
    d_offs = torch.zeros(max_lines + 1, dtype=torch.int64, device=dev)
    d_npos = torch.zeros(max_lines, dtype=torch.int64, device=dev)
    d_nlen = torch.zeros(max_lines, dtype=torch.int32, device=dev)
    wsb = lib.skh_parse_workspace_size(n, max_lines); ws_t, ws = aligned(wsb)
    d_res = torch.zeros(6, dtype=torch.int64, device=dev)
    st = torch.cuda.Stream(device=dev); sp = C.c_void_p(st.cuda_stream)
    torch.cuda.synchronize()

    def parse():
        lib.check(lib.skh_parse_dev(ctx, d_text.data_ptr(), n, capi.FASTQ if fastq else capi.FASTA, 1, d_seq.data_ptr(),
                                    d_qual.data_ptr(), padded, d_offs.data_ptr(), d_npos.data_ptr(), d_nlen.data_ptr(),
                                    max_lines, ws, wsb, max_lines, d_res.data_ptr(), sp))
    parse(); torch.cuda.synchronize()
    res = capi.ParseResult.from_buffer_copy(d_res.cpu().numpy().tobytes())
    if os.environ.get("PARSE_ONLY"):
        # kernel experiments whose builds may not parse correctly: the parse calls alone, for a kernel trace
        for _ in range(8): parse()
        torch.cuda.synchronize()
        print("   parse only: status %d nrec %d" % (res.status, res.nrec))
        os.remove(path)
        return
    assert res.status == 0 and res.nrec == nrec and res.seq_bytes == nrec * L, (res.status, res.nrec, res.seq_bytes)
    S = res.seq_bytes
    planes = 2 if fastq else 1

    cap = n + n // 8 + 4096
    d_out = torch.zeros(cap, dtype=torch.uint8, device=dev)
    fwsb = lib.skh_format_workspace_size(nrec); fws_t, fws = aligned(fwsb)
    d_fres = torch.zeros(2, dtype=torch.int64, device=dev)
    fmt = capi.SeqFormat(min_len=-1, max_len=-1, line_width=0 if fastq else 60, print_name=1, print_seq=1,
                         print_qual=1 if fastq else 0, plus_line=1, only_id=0, mark=b"@" if fastq else b">")

    torch.cuda.synchronize()      # the buffers above were zero-filled on the default stream

    def format_seq():
        lib.check(lib.skh_format_seq_dev(ctx, d_text.data_ptr(), d_npos.data_ptr(), d_nlen.data_ptr(), d_seq.data_ptr(),
                                         d_qual.data_ptr(), d_offs.data_ptr(), nrec, C.byref(fmt), d_out.data_ptr(), cap,
                                         fws, fwsb, d_fres.data_ptr(), sp))
    format_seq(); torch.cuda.synchronize()
    fres = capi.FormatResult.from_buffer_copy(d_fres.cpu().numpy().tobytes())
    assert fres.status == 0 and fres.out_bytes == n, (fres.status, fres.out_bytes, n)
    if not os.environ.get("SKIP_CHECK") and not bool(torch.equal(d_out[:n], d_text[:n])):
        bad = torch.nonzero(d_out[:n] != d_text[:n])[:, 0]
        i = int(bad[0])
        print("MISMATCH: %d bytes differ, first at %d" % (len(bad), i))
        print("  want", bytes(d_text[max(0, i - 40):i + 40].cpu().numpy().tobytes()))
        print("  got ", bytes(d_out[max(0, i - 40):i + 40].cpu().numpy().tobytes()))
        sys.exit(1)

    counts = torch.zeros(nrec * 4, dtype=torch.int64, device=dev)
    cls = lib.class_table(["G", "C", "A", "T"])
    torch.cuda.synchronize()
    lib.check(lib.skh_count_dev(ctx, d_seq.data_ptr(), d_offs.data_ptr(), nrec, S, cls, 4, counts.data_ptr(), sp))
    tf = capi.TabFormat()
    tf.ncols = 6; tf.only_id = 0; tf.print_seq = 0; tf.print_qual = 0
    kinds = [(capi.TAB_LENGTH, []), (capi.TAB_GC, [0, 1]), (capi.TAB_CONTENT, [2]), (capi.TAB_CONTENT, [0]),
             (capi.TAB_CONTENT, [1]), (capi.TAB_CONTENT, [3])]
    for i, (k, terms) in enumerate(kinds):
        tf.cols[i].kind = k; tf.cols[i].nterm = len(terms)
        for j, t in enumerate(terms): tf.cols[i].term[j] = t
    cptr = (C.c_void_p * 4)(counts.data_ptr(), None, None, None)
    ncl = (C.c_uint32 * 4)(4, 0, 0, 0)

    def dense_names():
        # experiment: what the rows would cost if the names stood one behind the other (a dense name plane) instead
        # of 340 bytes apart in the text -- the spans point into the start of the text, the bytes are not the names
        # (called behind the parse case, which writes the spans)
        d_npos.copy_(torch.cumsum(d_nlen.to(torch.int64), 0) - d_nlen.to(torch.int64))
        torch.cuda.synchronize()

    def format_tab():
        lib.check(lib.skh_format_tab_dev(ctx, d_text.data_ptr(), d_npos.data_ptr(), d_nlen.data_ptr(), d_seq.data_ptr(),
                                         d_qual.data_ptr(), d_offs.data_ptr(), nrec, cptr, ncl, C.byref(tf),
                                         d_out.data_ptr(), cap, fws, fwsb, d_fres.data_ptr(), sp))
    format_tab(); torch.cuda.synchronize()
    tab_bytes = capi.FormatResult.from_buffer_copy(d_fres.cpu().numpy().tobytes()).out_bytes
    name_bytes = int(d_nlen[:nrec].sum())

    cases = {
        "parse %s" % kind: (parse, n + planes * S + 20 * nrec),
        "format seq %s" % kind: (format_seq, name_bytes + planes * S + 20 * nrec + n),
        "format fx2tab -n -l -g -B A,G,C,T": (format_tab, name_bytes + 52 * nrec + tab_bytes),
    }
    print("== %s: %d records x %d, text %.2f GB" % (kind, nrec, L, n / 1e9))
    with torch.cuda.stream(st):
        for name, (fn, nbytes) in cases.items():
            if (os.environ.get("DENSE_NAMES") and name.startswith("format fx2tab")) or \
                    (os.environ.get("DENSE_NAMES") == "all" and name.startswith("format seq")):
                dense_names()
            for _ in range(2): fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(5): fn()
            e1.record(st)
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            print("   %-36s %8.3f ms  %7.1f GB/s algorithmic  %5.1f%% of 8 TB/s  %8.1f Gbases/s" %
                  (name, ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / 80, S / ms / 1e6))
    os.remove(path)


for k in os.environ.get("KINDS", "fq,fa").split(","):
    run(k)
lib.skh_destroy(ctx)
