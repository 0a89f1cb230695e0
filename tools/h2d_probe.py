#!/usr/bin/env python3
"""Raw host <-> device copy rates of this box (200 MB, pinned and pageable): what the staged encode pipeline can count on
(it needs ~6.7 GB/s of H2D for fp32 loader batches at 17 k images/s).   Usage: python tools/h2d_probe.py"""
import torch, time
n = 200 * 1024 * 1024
src = torch.empty(n, dtype=torch.uint8).pin_memory()
dst = torch.empty(n, dtype=torch.uint8, device="cuda")
pg = torch.empty(n, dtype=torch.uint8)
for name, s in (("pinned", src), ("pageable", pg)):
    for _ in range(2):
        dst.copy_(s, non_blocking=True); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        dst.copy_(s, non_blocking=True)
    torch.cuda.synchronize()
    print(name, "H2D %.1f GB/s" % (5 * n / (time.perf_counter() - t) / 1e9))
t = time.perf_counter()
for _ in range(5):
    src.copy_(dst, non_blocking=True)
torch.cuda.synchronize()
print("D2H pinned %.1f GB/s" % (5 * n / (time.perf_counter() - t) / 1e9))
