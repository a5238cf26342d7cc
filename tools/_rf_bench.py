import os, sys, numpy as np, torch, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seqkit_amd import capi
lib = capi.load(); ctx = lib.init(0)
dev = torch.device("cuda:0")
p = lambda t: C.c_void_p(t.data_ptr())
rng = np.random.default_rng(7)
shapes = {"trim": rng.integers(30, 152, 8_000_000), "r150": np.full(10_000_000, 150), "r100": np.full(10_000_000, 100), "r250": np.full(5_000_000, 250),
          "hp": np.clip(np.exp(rng.normal(np.log(100), 0.35, 8_000_000)), 39, 2354).astype(np.int64), "r75": np.full(12_000_000, 75), "r50": np.full(12_000_000, 50), "u350": rng.integers(280, 420, 4_000_000), "u700": rng.integers(560, 840, 2_000_000), "u1000": rng.integers(800, 1200, 1_500_000), "u2000": rng.integers(1600, 2400, 800_000), "u4000": rng.integers(3200, 4800, 400_000), "u6000": rng.integers(4800, 7200, 300_000), "ont": np.clip(np.exp(rng.normal(np.log(8000), 1.1, 200_000)), 1000, 1_000_000).astype(np.int64), "mix": np.concatenate([np.full(5_000_000, 150), np.full(200, 3_000_000)])}
out_line = []
for name in os.environ.get("WHICH", "trim,r150,r100,r250").split(","):
    lengths = shapes[name].astype(np.int64)
    offs_h = np.zeros(len(lengths) + 1, np.int64); np.cumsum(lengths, out=offs_h[1:])
    total, nrec = int(offs_h[-1]), len(lengths)
    seq = torch.randint(65, 85, ((total + 15) & ~15,), dtype=torch.uint8, device=dev)
    out = torch.empty_like(seq); offs = torch.from_numpy(offs_h).to(dev)
    lut = lib.pair_table(1)
    st = torch.cuda.Stream(device=dev); sp = C.c_void_p(st.cuda_stream)
    fn = lambda: lib.skh_revcomp_dev(ctx, p(seq), p(out), p(offs), nrec, total, lut, 1, sp)
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(60): fn()
    e1.record(st); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 60
    out_line.append("%s %.3f ms %.1f%%" % (name, ms, (2 * total + 8 * nrec) / ms / 1e6 / 80))
    del seq, out
print("  ".join(out_line))
