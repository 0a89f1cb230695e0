#!/usr/bin/env python3
"""How fast does the host stage a loader batch into page-locked memory?  One 64-image fp32 batch (25 MB, pageable) copied into
a pinned buffer by 1 / 2 / 4 threads (torch's copy_ releases the GIL), GB/s.  The staged encode pipeline needs ~6.7 GB/s at
17 k images/s.   Usage: python tools/host_copy_probe.py"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import torch

n = 64 * 3 * 256 * 128 * 4
srcs = [torch.randint(0, 255, (n,), dtype=torch.uint8) for _ in range(16)]   # 400 MB of distinct pageable batches
dst = torch.empty(n * 8, dtype=torch.uint8).pin_memory() if torch.cuda.is_available() else torch.empty(n * 8, dtype=torch.uint8)
print("cpus", os.cpu_count(), "torch threads", torch.get_num_threads())
for workers in (1, 2, 4, 8):
    pool = ThreadPoolExecutor(max_workers=workers)
    def copy(i):
        s = srcs[i % 16]
        d = dst[(i % 8) * n:(i % 8 + 1) * n]
        if workers == 1:
            d.copy_(s)
            return
        step = (n // workers + 4095) // 4096 * 4096
        futs = [pool.submit(lambda a, b: d[a:b].copy_(s[a:b]), o, min(n, o + step)) for o in range(0, n, step)]
        for f in futs:
            f.result()
    for i in range(8):
        copy(i)
    t = time.perf_counter()
    reps = 64
    for i in range(reps):
        copy(i)
    dt = time.perf_counter() - t
    print("%d thread(s): %.1f GB/s" % (workers, reps * n / dt / 1e9))
    pool.shutdown()

# the pipeline's stager is NOT the main thread: the same single copy_ from a worker thread, main thread idle / busy in torch
import threading

def worker(out):
    for i in range(8):
        dst[(i % 8) * n:(i % 8 + 1) * n].copy_(srcs[i % 16])
    t = time.perf_counter()
    for i in range(64):
        dst[(i % 8) * n:(i % 8 + 1) * n].copy_(srcs[i % 16])
    out.append(64 * n / (time.perf_counter() - t) / 1e9)

for busy in (False, True):
    res = []
    th = threading.Thread(target=worker, args=(res,))
    th.start()
    if busy and torch.cuda.is_available():
        a = torch.randn(2048, 2048, device="cuda")
        while th.is_alive():
            (a @ a).sum().item()
    th.join()
    print("copy_ from a worker thread, main thread %s: %.1f GB/s" % ("launching GPU work" if busy else "idle", res[0]))
