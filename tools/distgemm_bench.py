#!/usr/bin/env python3
"""feat x feat^T distance GEMM timings (north_star: 20k x 20k x 768): exact fp32 MFMA, one-pass fp16, 3-term split.
Usage: python tools/distgemm_bench.py [nq ng d [modes [full]]] ; with nq == ng the two operands are the SAME tensor (all-pairs
distances of one set: the fp16 kernels take their symmetric form) unless the fifth argument is "full" (a copy as the second
operand: the full computation).  MPREID_TUNE=gemm_stagger=<ticks> to try the start stagger."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mp-reid_amd")]
import torch  # noqa: E402
from mpreid import ops, synth  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
ng = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
d = int(sys.argv[3]) if len(sys.argv) > 3 else 768
modes = sys.argv[4].split(",") if len(sys.argv) > 4 else ["f16", "split3", "exact"]
f, _ = synth.clustered_features(max(nq, ng), d, 3.0, seed=1234)
ft = torch.from_numpy(f).cuda()
q, g = ft[:nq], ft[:ng]
full = len(sys.argv) > 5 and sys.argv[5] == "full"
if full:
    g = g.clone()
out = torch.empty((nq, ng), device="cuda")
flop = 2.0 * nq * ng * d
for name in modes:
    mode = {"f16": ops.GEMM_F16_FAST, "split3": ops.GEMM_F16_SPLIT3, "exact": ops.GEMM_F32_EXACT}[name]
    for _ in range(2):
        ops.euclidean_distance(q, g, mode=mode, out=out)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            ops.euclidean_distance(q, g, mode=mode, out=out)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 3)
    mult = 3 if name == "split3" else 1
    print(f"{name:6s} {nq}x{ng}x{d}: {best:.4f} ms  algorithmic {flop/best/1e9:.1f} TF  executed {mult*flop/best/1e9:.1f} TF  "
          f"store {4.0*nq*ng/best/1e6:.0f} GB/s  {'full' if full or nq != ng else 'same-tensor (symmetric form for f16 / split3)'}  tune={os.environ.get('MPREID_TUNE','')}", flush=True)
