#!/usr/bin/env python3
"""Host<->device copy rates for pinned blocks of several sizes, one direction and both at once."""
import time
import torch

dev = torch.device("cuda:0")
for mb in (8, 32, 128, 512):
    n = mb << 20
    h_in = torch.empty(n, dtype=torch.uint8).pin_memory()
    h_out = torch.empty(n, dtype=torch.uint8).pin_memory()
    h_in.fill_(65)
    d_a = torch.empty(n, dtype=torch.uint8, device=dev)
    d_b = torch.empty(n, dtype=torch.uint8, device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    reps = max(4, (2 << 30) // n)

    def run(h2d, d2h):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            if h2d:
                with torch.cuda.stream(s1):
                    d_a.copy_(h_in, non_blocking=True)
            if d2h:
                with torch.cuda.stream(s2):
                    h_out.copy_(d_b, non_blocking=True)
        torch.cuda.synchronize()
        return reps * n / (time.perf_counter() - t0) / 1e9

    run(True, True)
    print("%4d MiB blocks: H2D %.1f GB/s, D2H %.1f GB/s, both at once %.1f GB/s each way"
          % (mb, run(True, False), run(False, True), run(True, True)), flush=True)
