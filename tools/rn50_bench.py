#!/usr/bin/env python3
"""RN50 encoder throughput on synthetic images (random-init weights): images/s and GFLOP/s of the conv trunk.
Usage: python tools/rn50_bench.py [--batch 256] [--steps 5]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mp-reid_amd")]
import torch  # noqa: E402
from mpreid import ops, synth  # noqa: E402

GFLOP_PER_IMG = 9.32 + 2.16 + 0.03   # conv trunk + K/V projections of the attention pool + q / c_proj (256x128 input)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--precision", default="fp16", choices=("fp16", "split", "fp32"))
    a = ap.parse_args()
    enc = ops.Rn50Encoder(synth.RN50, synth.rn50_state_dict(synth.RN50, seed=11), (256, 128), precision=a.precision)
    img = torch.from_numpy(synth.synthetic_images(min(a.batch, 64), 256, 128, seed=1)).cuda()
    img = img.repeat((a.batch + img.shape[0] - 1) // img.shape[0], 1, 1, 1)[:a.batch].contiguous()
    out = torch.empty((a.batch, enc.feat_dim), device="cuda")
    enc(img, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        enc(img, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(f"rn50 {a.precision} batch {a.batch}: {dt*1e3:.2f} ms/batch  {a.batch/dt:.0f} img/s  {a.batch*GFLOP_PER_IMG/dt/1e3:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
