#!/usr/bin/env python3
"""Per-tile phase timeline of the two-workgroups-per-CU symmetric distance kernel (ablation build only:
MPREID_ABLATION=1 python mp-reid_amd/mpreid/build.py -> tools/ablation_lib/libmpreid_hip_abl.so; run with
MPREID_LIB=tools/ablation_lib/libmpreid_hip_abl.so MPREID_ALLOW_ABLATION=1).
Stamps of the 100 MHz counter per workgroup and tile: 0 operands of stage 0 landed, 1 k-loop done, 2 epilogue barrier passed (ring free),
3 stores issued, 4 stores drained.  Prints the averages and the interleaving of the two workgroups of a few CUs
(blocks b and b + 256).   Usage: python tools/p2_stamps.py [n d]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mp-reid_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
from mpreid import ops, synth  # noqa: E402

f, _ = synth.clustered_features(n, d, 3.0, seed=1234)
ft = torch.from_numpy(f).cuda()
out = torch.empty((n, n), device="cuda")
stamps = torch.zeros((512, 32, 8), dtype=torch.int64, device="cuda")
os.environ["MPREID_GEMM_STAMPS"] = "%x" % stamps.data_ptr()
for _ in range(3):
    stamps.zero_()
    ops.euclidean_distance(ft, ft, mode=ops.GEMM_F16_FAST, out=out)
torch.cuda.synchronize()
s = stamps.cpu().numpy().astype(np.int64)
ok = s[:, :, 0] > 0
t0 = s[:, :, 0][ok].min()
us = lambda x: x / 100.0
print("tiles per workgroup %d-%d; span %.0f us" % (ok.sum(1).min(), ok.sum(1).max(), us(s[:, :, 4].max() - t0)))
for name, a, b in (("k-loop", 0, 1), ("tables+barrier", 1, 2), ("store issue", 2, 3), ("drain", 3, 4)):
    v = us((s[:, :, b] - s[:, :, a])[ok])
    print("%-12s mean %.2f us  p10 %.2f  p90 %.2f" % (name, v.mean(), np.percentile(v, 10), np.percentile(v, 90)))
nx = us((s[:, 1:, 0] - s[:, :-1, 4])[ok[:, 1:]])
print("%-12s mean %.2f us" % ("restart", nx.mean()))
for b in (0, 9, 100):
    print("CU of blocks %d and %d (us from the first stamp):" % (b, b + 256))
    for w in (b, b + 256):
        line = []
        for t in range(min(6, ok[w].sum())):
            line.append("k %.1f-%.1f st %.1f-%.1f-%.1f" % tuple(us(s[w, t, i] - t0) for i in (0, 1, 2, 3, 4)))
        print("   wg %3d: " % w + " | ".join(line))
