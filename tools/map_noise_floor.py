"""How much do mAP / Rank-1 of the image->mAP parity sets (tests/test_gpu_map_parity.py: SETS) move under the fp32
rounding noise of the REFERENCE itself?  oracle fp32 features vs the same graph evaluated in fp64 (CPU only).
Usage: python tools/map_noise_floor.py [spread|degenerate]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mp-reid_amd")]
import numpy as np
import torch
from mpreid import synth
from oracle import oracle as orc

SETS = {"spread": (128, 8, 0.4, 0.05), "degenerate": (128, 16, 0.55, 0.02)}   # tests/test_gpu_map_parity.py
name = sys.argv[1] if len(sys.argv) > 1 else "spread"
n_ids, per_id, beta, std = SETS[name]
x, pid = synth.identity_images(n_ids, per_id, beta)
sd = synth.vit_state_dict(synth.VIT_B16, seed=7, std=std)
print("set", name, SETS[name], flush=True)
n = len(pid); nq = n // 5
t0 = time.time()
f32 = np.concatenate([orc.vit_features(sd, synth.VIT_B16, x[s:s + 64]) for s in range(0, n, 64)])
print("fp32 features", time.time() - t0, flush=True)
t0 = time.time()
f64 = np.concatenate([orc.vit_features(sd, synth.VIT_B16, x[s:s + 64], dtype="float64") for s in range(0, n, 64)])
print("fp64 features", time.time() - t0, flush=True)
print("rel-L2 fp32 vs fp64", np.linalg.norm(f32 - f64) / np.linalg.norm(f64))

def ev64(f):
    f = f / np.linalg.norm(f, axis=1, keepdims=True)
    q, g = f[:nq], f[nq:]
    d = (q * q).sum(1)[:, None] + (g * g).sum(1)[None, :] - 2 * q @ g.T
    return orc.eval_func(d.astype(np.float64), pid[:nq], pid[nq:]), d

(c64, m64), d64 = ev64(f64.astype(np.float64))
fo = orc.l2_normalize(f32)
c32, m32 = orc.eval_func(orc.euclidean_distance(fo[:nq], fo[nq:]), pid[:nq], pid[nq:])
print("fp64: mAP %.6f R1 %.6f | oracle fp32: mAP %.6f R1 %.6f | d mAP %.2e dR1 %.2e" % (m64, c64[0], m32, c32[0], abs(m64 - m32), abs(c64[0] - c32[0])))
# rank-1 margins of the exact distances: how many queries have a top-2 gap below k * 1e-6?
srt = np.sort(d64, axis=1)
gap = srt[:, 1] - srt[:, 0]
print("top-2 gap quantiles", np.quantile(gap, [0, 0.01, 0.05, 0.5]), "typical distance", np.median(srt[:, 0]))
rng = np.random.default_rng(0)
for rel in (1e-6, 1.4e-6, 3e-6):
    flips = []
    for t in range(10):
        fp = f64 * (1 + rel * np.sqrt(3) * rng.uniform(-1, 1, f64.shape))  # relative noise of rms `rel`
        (c, m), _ = ev64(fp)
        flips.append((abs(m - m64), abs(c[0] - c64[0])))
    print(f"noise {rel:.1e}: dmAP max {max(a for a, _ in flips):.2e} mean {np.mean([a for a, _ in flips]):.2e}; dR1 nonzero in {sum(b > 0 for _, b in flips)}/10 trials")
