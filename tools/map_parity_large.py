#!/usr/bin/env python3
"""Image -> mAP parity of the encoder precisions on a LARGER set than tests/test_gpu_map_parity.py's (one-off measurement
for DESIGN.md; the oracle encode runs on the host cores: ~25 images/s).
Usage: python tools/map_parity_large.py [n_ids per_id beta std]   (default 512 ids x 8 images, spread geometry)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mp-reid_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
from mpreid import ops, synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

n_ids = int(sys.argv[1]) if len(sys.argv) > 1 else 512
per_id = int(sys.argv[2]) if len(sys.argv) > 2 else 8
beta = float(sys.argv[3]) if len(sys.argv) > 3 else 0.4
std = float(sys.argv[4]) if len(sys.argv) > 4 else 0.05
torch.set_num_threads(min(torch.get_num_threads(), 32))
x, pid = synth.identity_images(n_ids, per_id, beta)
sd = synth.vit_state_dict(synth.VIT_B16, seed=7, std=std)
n = len(pid)
nq = n // 5
t0 = time.time()
f_or = np.concatenate([orc.vit_features(sd, synth.VIT_B16, x[s:s + 64]) for s in range(0, n, 64)])
print(f"oracle encode of {n} images: {time.time() - t0:.0f} s", flush=True)
fo = orc.l2_normalize(f_or)
ref = {}
for rr in (False, True):
    d = orc.re_ranking(fo[:nq], fo[nq:], 50, 15, 0.3) if rr else orc.euclidean_distance(fo[:nq], fo[nq:])
    ref[rr] = orc.eval_func(d, pid[:nq], pid[nq:])
    print(f"oracle rerank={rr}: mAP {ref[rr][1]:.6f} R1 {ref[rr][0][0]:.6f}", flush=True)
print("median normalised distance %.4f" % float(np.median(orc.euclidean_distance(fo[:nq], fo[nq:]))))
for prec in ("split", "fp32", "fp16"):
    enc = ops.VitEncoder(synth.VIT_B16, sd, (256, 128), precision=prec)
    f = torch.empty((n, enc.feat_dim), device="cuda")
    for s in range(0, n, 508):
        enc(torch.from_numpy(x[s:s + 508]), out=f[s:s + 508])
    rel = float(np.linalg.norm(f.cpu().numpy() - f_or) / np.linalg.norm(f_or))
    fn = ops.l2_normalize(f)
    for rr in (False, True):
        d = ops.re_ranking(fn[:nq], fn[nq:], 50, 15, 0.3)[0] if rr else ops.euclidean_distance(fn[:nq], fn[nq:])
        cmc, mAP = orc.eval_func(d.cpu().numpy(), pid[:nq], pid[nq:])
        print(f"{prec:5s} rerank={rr}: feat rel-L2 {rel:.2e}  |dmAP| {abs(mAP - ref[rr][1]):.2e}  |dR1| "
              f"{abs(float(cmc[0]) - float(ref[rr][0][0])):.2e} ({round(abs(float(cmc[0]) - float(ref[rr][0][0])) * nq)} of {nq} queries)  "
              f"max |dCMC| {float(np.abs(cmc - ref[rr][0]).max()):.2e}", flush=True)
    del enc
