#!/usr/bin/env python3
"""Run-to-run determinism of the encoders under the bench's two-stream concurrency: the same batch is encoded
repeatedly on two streams at once (two encoder handles, own workspaces) and every output must equal the first bit
for bit (rare LDS / vmcnt races show up as flipped low bits).  Usage: python tools/encoder_determinism.py [--reps 12]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mp-reid_amd")]
import torch  # noqa: E402
from mpreid import ops, synth  # noqa: E402


def check(name, make, img, reps):
    e0 = make("a")
    e1 = make("b")
    side = torch.cuda.Stream()
    ref = e0(img).clone()
    torch.cuda.synchronize()
    bad = 0
    for r in range(reps):
        o0 = torch.empty_like(ref)
        o1 = torch.empty_like(ref)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            e1(img, out=o1)
        e0(img, out=o0)
        torch.cuda.synchronize()
        bad += int(not torch.equal(o0, ref)) + int(not torch.equal(o1, ref))
    print(f"{name}: {2 * reps - bad}/{2 * reps} runs bit-identical")
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=12)
    a = ap.parse_args()
    base = torch.from_numpy(synth.synthetic_images(64, 256, 128, seed=3)).cuda()
    img = base.repeat(8, 1, 1, 1)[:508].contiguous()
    sd = synth.vit_state_dict(synth.VIT_B16, seed=7)
    bad = check("ViT-B/16 fp16 mode, batch 508", lambda t: ops.VitEncoder(synth.VIT_B16, sd, (256, 128), ws_tag="det_" + t, precision="fp16"), img, a.reps)
    bad += check("ViT-B/16 split mode (default), batch 508",
                 lambda t: ops.VitEncoder(synth.VIT_B16, sd, (256, 128), ws_tag="dets_" + t, precision="split"), img, a.reps)
    sdr = synth.rn50_state_dict(synth.RN50, seed=11)
    bad += check("RN50 batch 256", lambda t: ops.Rn50Encoder(synth.RN50, sdr, (256, 128), ws_tag="detr_" + t, precision="fp16"),
                 img[:256].contiguous(), a.reps)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
