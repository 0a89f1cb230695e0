"""images/s of the ViT-B/16 encoder per precision mode (fp16 | split | fp32), inputs resident in HBM, one stream.
   python tools/encoder_modes_bench.py [batch] [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mp-reid_amd")]

import torch  # noqa: E402
from mpreid import _lib, ops, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 508
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
modes = sys.argv[3].split(",") if len(sys.argv) > 3 else ["fp16", "split", "fp32"]
dev = _lib.require_gpu()
sd = synth.vit_state_dict(synth.VIT_B16, seed=7)
g = torch.Generator(device=dev)
g.manual_seed(1)
img = torch.randn((B, 3, 256, 128), generator=g, device=dev).clamp_(-1, 1)
L = _lib.load()
for prec in modes:
    enc = ops.VitEncoder(synth.VIT_B16, sd, (256, 128), precision=prec)
    out = torch.empty((B, enc.feat_dim), device=dev)
    enc(img, out=out)
    torch.cuda.synchronize()
    n = reps if prec != "fp32" else max(1, reps // 3)
    L.mpreid_profile_reset()
    L.mpreid_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(n):
        enc(img, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    L.mpreid_profile_enable(0)
    ents = (_lib.ProfileEntry * 16)()
    k = L.mpreid_profile_query(ents, 16)
    cls = "; ".join(f"{_lib.GEMM_EPILOGUE_NAMES.get(ents[i].epilogue)} N{ents[i].n} K{ents[i].k}: "
                    f"{ents[i].total_ms / max(ents[i].launches, 1) * 1e3:.0f} us {ents[i].flops_total / max(ents[i].total_ms, 1e-9) / 1e9:.0f} TF"
                    for i in range(min(k, 5)))
    print(f"{prec}: {B / dt:.0f} images/s ({dt * 1e3:.2f} ms per batch of {B})  | {cls}", flush=True)
    del enc
