#!/usr/bin/env python3
"""Re-rank stage timings (hipEvents inside the library) on clustered synthetic features."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mp-reid_amd")]
import torch  # noqa: E402
from mpreid import ops, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else N // 5
d = int(sys.argv[3]) if len(sys.argv) > 3 else 768
f, _ = synth.clustered_features(N, d, 3.0, seed=1234)
ft = torch.from_numpy(f).cuda()
ops.re_ranking(ft[:nq], ft[nq:], 50, 15, 0.3)
best = None
for _ in range(3):
    _, st = ops.re_ranking(ft[:nq], ft[nq:], 50, 15, 0.3, timing=True)
    if best is None or st["ms_total"] < best["ms_total"]:
        best = st
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in best.items()})
