#!/usr/bin/env python3
"""Two encoder forwards of one batch (profiling target for rocprofv3 --pmc)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mp-reid_amd")]
import torch  # noqa: E402
from mpreid import ops, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 508
enc = ops.VitEncoder(synth.VIT_B16, synth.vit_state_dict(synth.VIT_B16, seed=7), (256, 128))
img = torch.randn((B, 3, 256, 128), device="cuda").clamp_(-1, 1)
for _ in range(2):
    out = enc(img)
torch.cuda.synchronize()
print(float(out.abs().mean()))
