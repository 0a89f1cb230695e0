import os, sys, time
sys.path[:0]=['/root/repo','/root/repo/mp-reid_amd']
import numpy as np, torch
from mpreid import synth
from oracle import oracle as orc
print("cpu_count", os.cpu_count(), "torch default threads", torch.get_num_threads(), flush=True)
sd = synth.vit_state_dict(synth.VIT_B16, seed=7, std=0.05)
x = synth.synthetic_images(128, 256, 128, seed=1)
orc.vit_features(sd, synth.VIT_B16, x[:8])
for th in (16, 32, 48, 64, 96):
    if th > (os.cpu_count() or 1): break
    torch.set_num_threads(th)
    for bs in (64, 128):
        t0=time.perf_counter()
        for s in range(0,128,bs): orc.vit_features(sd, synth.VIT_B16, x[s:s+bs])
        print(th, "threads, batch", bs, ":", round(128/(time.perf_counter()-t0),1), "img/s", flush=True)
