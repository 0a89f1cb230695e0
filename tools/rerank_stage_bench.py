#!/usr/bin/env python3
"""Stage times of the single-call re-ranking on SURVEY §8d clustered features generated on the device.
Usage: python tools/rerank_stage_bench.py [N nq D sigma] ; prints the stats dictionary of the best of 3 runs."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mp-reid_amd")]
import torch  # noqa: E402
from mpreid import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else N // 5
d = int(sys.argv[3]) if len(sys.argv) > 3 else 768
sigma = float(sys.argv[4]) if len(sys.argv) > 4 else 3.0
g = torch.Generator(device="cuda")
g.manual_seed(4321)
cent = torch.randn((N // 20, d), generator=g, device="cuda")
pid = torch.randint(0, N // 20, (N,), generator=g, device="cuda")
f = ops.l2_normalize(cent[pid] + sigma * torch.randn((N, d), generator=g, device="cuda"))
if os.environ.get("BENCH_SORT_BY_PID"):   # locality experiment: rows of one identity adjacent (as in a file-name-sorted dataset)
    f = f[torch.argsort(pid, stable=True)].contiguous()
best = None
for _ in range(4):
    out, st = ops.re_ranking(f[:nq], f[nq:], 50, 15, 0.3, timing=True)
    if best is None or st["ms_total"] < best["ms_total"]:
        best = st
print(f"N={N} nq={nq} D={d}: total {best['ms_total']:.2f} ms  " +
      " ".join(f"{k[3:]}={v:.2f}" for k, v in best.items() if k.startswith("ms_") and k != "ms_total"),
      f"pairs={best['jaccard_pairs']:.3g} vqe_nnz={best['vqe_nnz']:.3g} sha={hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:12]}")
