#!/usr/bin/env python3
"""eval_rank_kernel (csrc/evalrank.hip) at Market-1501 shape: alone, beside the D2H copy of the same matrix (round 5's
order inside R1_mAP_eval.compute()), and the whole of compute() in both orders.

    python tools/evalrank_bench.py [nq ng]

Round 5's trace (profiles/r05_bench_kernel_stats.csv) showed the kernel at 0.22 ms min / 5.39 ms average: this script
separates the kernel's own time from what the concurrent blit copy does to it."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "mp-reid_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from mpreid import _lib, ops, synth  # noqa: E402
from utils.metrics import R1_mAP_eval  # noqa: E402


def main():
    nq, ng = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3368, 15913)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    L = _lib.load()
    rng = np.random.default_rng(1234)
    pids = rng.integers(0, max(ng // 21, 1), size=nq + ng).astype(np.int64)
    dist = torch.randn((nq, ng), device=dev)
    qp, gp = torch.from_numpy(pids[:nq]).to(dev), torch.from_numpy(pids[nq:]).to(dev)
    rcap = int(np.unique(pids[nq:], return_counts=True)[1].max())
    pos = torch.empty((nq, rcap), dtype=torch.int32, device=dev)
    cnt = torch.empty(nq, dtype=torch.int32, device=dev)

    def launch():
        _lib.check(L.mpreid_eval_rank_positions(C.c_void_p(dist.data_ptr()), dist.stride(0), nq, ng, C.c_void_p(qp.data_ptr()),
                                                C.c_void_p(gp.data_ptr()), rcap, C.c_void_p(pos.data_ptr()),
                                                C.c_void_p(cnt.data_ptr()), _lib.stream_ptr()), "eval_rank")

    def timed(fn, reps=10):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return min(ts), sum(ts) / len(ts), max(ts)

    # host check of the positions (stable argsort of a few rows)
    launch()
    torch.cuda.synchronize()
    p_h, c_h = pos.cpu().numpy(), cnt.cpu().numpy()
    d_h = dist[:8].cpu().numpy()
    for r in range(8):
        order = np.argsort(d_h[r], kind="stable")
        want = np.nonzero(pids[nq:][order] == pids[r])[0]
        assert c_h[r] == want.size and np.array_equal(p_h[r, :want.size], want), r
    print(f"shape {nq} x {ng}, rcap {rcap}: positions of rows 0-7 equal a stable argsort")
    alone = timed(launch)
    print("kernel alone            min/avg/max ms: %.3f %.3f %.3f  (%.0f GB/s of 4*nq*ng)" % (*alone, 4.0 * nq * ng / alone[1] / 1e6))
    side = torch.cuda.Stream(device=dev)
    host = torch.empty((nq, ng), dtype=torch.float32, pin_memory=True)

    def beside():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            host.copy_(dist, non_blocking=True)
        launch()

    b = timed(beside)
    torch.cuda.synchronize()
    print("kernel beside the D2H   min/avg/max ms: %.3f %.3f %.3f" % b)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        e0.record()
        host.copy_(dist, non_blocking=True)
        e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print("D2H of the matrix alone: %.3f ms (%.1f GB/s)" % (ms, 4.0 * nq * ng / ms / 1e6))
    # the whole of compute()
    f, _ = synth.clustered_features(nq + ng, 1280, 3.5, seed=2)
    ft = torch.from_numpy(f).to(dev)
    for mode in ("behind", "overlap", "behind", "overlap"):
        os.environ["MPREID_EVAL_D2H"] = mode
        ev = R1_mAP_eval(nq, feat_norm=True)
        ts = []
        for it in range(6):
            ev.reset()
            for s in range(0, nq + ng, 64):
                ev.update((ft[s:s + 64], pids[s:s + 64], np.zeros(min(64, nq + ng - s), np.int64)))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with open(os.devnull, "w") as nul:
                so, sys.stdout = sys.stdout, nul
                try:
                    out = ev.compute()
                finally:
                    sys.stdout = so
            ts.append((time.perf_counter() - t0) * 1e3)
        print("compute() with the D2H %-8s: min %.2f ms, median %.2f ms (mAP %.6f)" % (mode, min(ts[1:]), sorted(ts[1:])[len(ts[1:]) // 2], out[1]))


if __name__ == "__main__":
    main()
