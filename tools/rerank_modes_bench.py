#!/usr/bin/env python3
"""Re-ranking in the bit-parity mode (RERANK_SPARSE) and with the blend term's distance rows from the fp16 matrix cores
(RERANK_SPARSE_SPLIT3): total / query-row times and the largest difference of the outputs, at N = 20 000, 100 000 and the
MSMT17 shape.  Usage: python tools/rerank_modes_bench.py"""
import os, sys
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "mp-reid_amd")]
import torch
from mpreid import ops
for (N, nq, d, sg) in ((20000, 4000, 768, 3.0), (100000, 20000, 768, 3.0), (93820, 11659, 1280, 3.5)):
    g = torch.Generator(device="cuda"); g.manual_seed(4321)
    cent = torch.randn((N // 20, d), generator=g, device="cuda")
    pid = torch.randint(0, N // 20, (N,), generator=g, device="cuda")
    f = ops.l2_normalize(cent[pid] + sg * torch.randn((N, d), generator=g, device="cuda"))
    res = {}
    for name, algo in (("exact", ops.RERANK_SPARSE), ("split3", ops.RERANK_SPARSE_SPLIT3)):
        best = None
        for _ in range(3):
            out, st = ops.re_ranking(f[:nq], f[nq:], 50, 15, 0.3, timing=True, algo=algo)
            if best is None or st["ms_total"] < best["ms_total"]: best = st
        res[name] = (out, best)
    dmax = float((res["exact"][0] - res["split3"][0]).abs().max())
    print(N, {k: round(v[1]["ms_total"], 2) for k, v in res.items()}, "dq", {k: round(v[1]["ms_dq"], 2) for k, v in res.items()}, "max |delta|", dmax)
    del res, f; ops.release_workspaces(); torch.cuda.empty_cache()
