#!/usr/bin/env python3
"""ViT-B/16 encoder throughput vs batch size on one stream (what the drop-in do_inference sees with
TEST.IMS_PER_BATCH = batch).  Usage: python tools/vit_batch_sweep.py [batches...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mp-reid_amd")]
import torch  # noqa: E402
from mpreid import ops, synth  # noqa: E402

batches = [int(x) for x in sys.argv[1:]] or [32, 64, 128, 256, 508, 1016]
enc = ops.VitEncoder(synth.VIT_B16, synth.vit_state_dict(synth.VIT_B16, seed=7), (256, 128))
base = torch.from_numpy(synth.synthetic_images(64, 256, 128, seed=1)).cuda()
for b in batches:
    img = base.repeat((b + 63) // 64, 1, 1, 1)[:b].contiguous()
    out = torch.empty((b, enc.feat_dim), device="cuda")
    enc(img, out=out)
    torch.cuda.synchronize()
    reps = max(2, 2048 // b)
    t0 = time.perf_counter()
    for _ in range(reps):
        enc(img, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"batch {b:5d}: {dt*1e3:8.3f} ms  {b/dt:9.0f} img/s  {b*21.12/dt/1e3:7.1f} TFLOP/s")
