#!/usr/bin/env python3
"""Image -> mAP parity probe (north_star: mAP / Rank-1 within 1e-4 of the fp32 CPU path): synthetic identity images
through (a) the HIP pipeline (fp16-MFMA ViT-B/16 -> normalise -> exact distance -> eval) and (b) the fp32 oracle
pipeline.  Prints both mAP / Rank-1 and the feature error.  GPU box only."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mp-reid_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mpreid import ops, synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n-ids", type=int, default=128)
ap.add_argument("--per", type=int, default=16)
ap.add_argument("--betas", type=str, default="0.55")
ap.add_argument("--oracle", type=int, default=1)
ap.add_argument("--rerank", type=int, default=0)
a = ap.parse_args()
sd = synth.vit_state_dict(synth.VIT_B16, seed=7)
enc = ops.VitEncoder(synth.VIT_B16, sd, (256, 128))
torch.set_num_threads(min(os.cpu_count() or 1, 32))
for beta in [float(b) for b in a.betas.split(",")]:
    x, pid = synth.identity_images(a.n_ids, a.per, beta)
    n = len(pid)
    nq = n // 5
    f_hip = torch.empty((n, enc.feat_dim), device="cuda")
    for s in range(0, n, 508):
        enc(torch.from_numpy(x[s:s + 508]), out=f_hip[s:s + 508])
    fn = ops.l2_normalize(f_hip)
    d_hip = (ops.re_ranking(fn[:nq], fn[nq:], 50, 15, 0.3)[0] if a.rerank else ops.euclidean_distance(fn[:nq], fn[nq:])).cpu().numpy()
    cmc_h, map_h = orc.eval_func(d_hip, pid[:nq], pid[nq:])
    line = f"beta {beta}: HIP mAP {map_h:.6f} R1 {cmc_h[0]:.6f}"
    if a.oracle:
        t0 = time.time()
        f_or = np.concatenate([orc.vit_features(sd, synth.VIT_B16, x[s:s + 64]) for s in range(0, n, 64)])
        t_or = time.time() - t0
        fo = orc.l2_normalize(f_or)
        d_or = orc.re_ranking(fo[:nq], fo[nq:], 50, 15, 0.3) if a.rerank else orc.euclidean_distance(fo[:nq], fo[nq:])
        cmc_o, map_o = orc.eval_func(d_or, pid[:nq], pid[nq:])
        rel = np.linalg.norm(f_hip.cpu().numpy() - f_or) / np.linalg.norm(f_or)
        reln = np.linalg.norm(fn.cpu().numpy() - fo, axis=1).max()
        line += (f" | oracle fp32 mAP {map_o:.6f} R1 {cmc_o[0]:.6f} ({t_or:.0f} s) | dmAP {abs(map_h - map_o):.2e} "
                 f"dR1 {abs(cmc_h[0] - cmc_o[0]):.2e} max dCMC {np.abs(cmc_h - cmc_o).max():.2e} | feat rel-L2 {rel:.2e} "
                 f"max row err (normalised) {reln:.2e} | dist range {d_or.min():.4f}..{d_or.max():.4f}")
    print(line, flush=True)
