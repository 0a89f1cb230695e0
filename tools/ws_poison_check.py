#!/usr/bin/env python3
"""Does an encoder read workspace memory it has not written?  Run a batch, poison the cached workspace (NaN bit pattern,
then a large finite pattern), run again: the features must be bit-identical.  Usage: python tools/ws_poison_check.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mp-reid_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
from mpreid import ops, synth  # noqa: E402

bad = 0
sd = synth.vit_state_dict(synth.VIT_B16, seed=7, std=0.02)
cases = [("vit split", lambda: ops.VitEncoder(synth.VIT_B16, sd, (256, 128), precision="split")),
         ("vit fp16", lambda: ops.VitEncoder(synth.VIT_B16, sd, (256, 128), precision="fp16")),
         ("rn50 split", lambda: ops.Rn50Encoder(synth.RN50, synth.rn50_state_dict(synth.RN50, seed=11), (256, 128), precision="split"))]
for name, make in cases:
    for n in (508, 16, 37):
        x = torch.from_numpy(synth.synthetic_images(min(n, 64), 256, 128, seed=n)).cuda()
        x = x.repeat((n + x.shape[0] - 1) // x.shape[0], 1, 1, 1)[:n].contiguous()
        x = x + 0.01 * torch.randn(x.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(n))
        enc = make()
        ref = enc(x).clone()
        torch.cuda.synchronize()
        for pat in (0xFF, 0x7B, 0x00):   # NaN halves / floats, large finite values, zeros
            for key, buf in list(ops._ws_cache.items()):
                buf.fill_(pat)
            got = enc(x).clone()
            torch.cuda.synchronize()
            same = torch.equal(ref, got)
            if not same:
                bad += 1
                d = (ref - got).abs()
                print(f"{name} n={n} poison=0x{pat:02X}: DIFFERENT  max abs {float(d.max()):.3e} rows affected {int((d.max(dim=1).values > 0).sum())} "
                      f"finite={bool(torch.isfinite(got).all())}", flush=True)
        print(f"{name} n={n}: done", flush=True)
        del enc
        ops.release_workspaces()
print("FAIL" if bad else "OK: no encoder reads workspace bytes it did not write")
sys.exit(1 if bad else 0)
