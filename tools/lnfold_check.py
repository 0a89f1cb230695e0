#!/usr/bin/env python3
"""Folded-LayerNorm split mode against the plain split mode and the fp32 oracle: accuracy, batch independence (the same
images through the 128 x 128 kernel in a small batch and the persistent kernel in a large one), images/s.
Usage: python tools/lnfold_check.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mp-reid_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
from mpreid import ops, synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

small = dict(h_res=4, w_res=2, patch=16, stride=16, width=128, layers=2, heads=2, out_dim=64)
for name, cfg, hw, std, jit in (("small", small, (64, 32), 0.05, 0.1), ("b16", synth.VIT_B16, (256, 128), 0.05, 0.05)):
    sd = synth.vit_state_dict(cfg, seed=7, std=std, ln_jitter=jit)
    imgs = synth.synthetic_images(6, hw[0], hw[1], seed=3)
    want = orc.vit_features(sd, cfg, imgs)
    for fold in (False, True):
        for cls_last in (True, False):
            enc = ops.VitEncoder(cfg, sd, hw, precision="split", ln_fold=fold, cls_only_last=cls_last)
            f = enc(torch.from_numpy(imgs)).cpu().numpy()
            print(f"{name} fold={fold} cls_only_last={cls_last}: rel-L2 vs oracle {np.linalg.norm(f - want) / np.linalg.norm(want):.2e}  "
                  f"max abs {np.abs(f - want).max():.2e}", flush=True)
sd = synth.vit_state_dict(synth.VIT_B16, seed=7, std=0.05)
enc = ops.VitEncoder(synth.VIT_B16, sd, (256, 128), precision="split", ln_fold=True)
imgs = synth.synthetic_images(6, 256, 128, seed=3)
f_small = enc(torch.from_numpy(imgs)).cpu().numpy()
big = synth.synthetic_images(508, 256, 128, seed=9)
big[100:106] = imgs
f_big = enc(torch.from_numpy(big)).cpu().numpy()
print("batch independence (6 images alone vs inside 508):", "bit-identical" if np.array_equal(f_small, f_big[100:106]) else
      f"DIFFERENT, max abs {np.abs(f_small - f_big[100:106]).max():.3e}")
f_big2 = enc(torch.from_numpy(big)).cpu().numpy()
print("run-to-run:", "bit-identical" if np.array_equal(f_big, f_big2) else "DIFFERENT")
x = torch.from_numpy(big).cuda()
for fold in (False, True):
    e = ops.VitEncoder(synth.VIT_B16, sd, (256, 128), precision="split", ln_fold=fold)
    out = torch.empty((508, e.feat_dim), device="cuda")
    e(x, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        e(x, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 8
    print(f"fold={fold}: {508 / dt:.0f} images/s ({dt * 1e3:.2f} ms per batch of 508)", flush=True)
from mpreid import _lib  # noqa: E402
L = _lib.load()
for fold in (False, True):
    e = ops.VitEncoder(synth.VIT_B16, sd, (256, 128), precision="split", ln_fold=fold)
    out = torch.empty((508, e.feat_dim), device="cuda")
    e(x, out=out)
    torch.cuda.synchronize()
    L.mpreid_profile_reset()
    L.mpreid_profile_enable(1)
    for _ in range(3):
        e(x, out=out)
    torch.cuda.synchronize()
    L.mpreid_profile_enable(0)
    ents = (_lib.ProfileEntry * 16)()
    k = L.mpreid_profile_query(ents, 16)
    print(f"fold={fold}: " + "; ".join(f"{_lib.GEMM_EPILOGUE_NAMES.get(ents[i].epilogue)} M{ents[i].m} N{ents[i].n} K{ents[i].k}: "
                                       f"{ents[i].total_ms / max(ents[i].launches, 1) * 1e3:.0f} us x{ents[i].launches}"
                                       for i in range(k)), flush=True)
