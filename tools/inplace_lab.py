#!/usr/bin/env python3
"""What one plane instead of two is worth to the memory system (GPU only): the complement-only kernel
(every 16-byte chunk read and written once, no reversal) out of place and in place, on planes of
several sizes."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import seqkit_amd

dev = torch.device("cuda", 0)
lib = seqkit_amd.load(); ctx = lib.init(0)
p = lambda t: C.c_void_p(t.data_ptr())
lut = lib.pair_table(1)
st = torch.cuda.Stream(device=dev); sp = C.c_void_p(st.cuda_stream)
for total in (1_500_000_000, 3_000_000_000, 15_000_000_000):
    seq = torch.randint(65, 85, (total,), dtype=torch.uint8, device=dev)
    out = torch.empty_like(seq)
    offs = torch.tensor([0, total], dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    ops = {
        "complement, two planes": lambda: lib.skh_revcomp_dev(ctx, p(seq), p(out), p(offs), 1, total, lut, 0, sp),
        "complement, in place": lambda: lib.skh_revcomp_inplace_dev(ctx, p(seq), None, p(offs), 1, total, lut, 0, sp),
    }
    for name, fn in ops.items():
        for _ in range(2): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(10): fn()
        e1.record(st)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("%5.1f GB  %-24s %8.3f ms %7.1f GB/s %5.1f%% of 8 TB/s" % (total / 1e9, name, ms, 2 * total / ms / 1e6, 2 * total / ms / 1e6 / 80))
    del seq, out
    torch.cuda.empty_cache()
lib.skh_destroy(ctx)
