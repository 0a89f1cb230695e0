"""Image -> mAP parity (north_star: "matching the reference CPU path's mAP / Rank-1 within 1e-4").

2048 synthetic images with identity structure (128 identities x 16 images, mpreid.synth.identity_images, beta chosen so
that the Euclidean mAP is ~0.55: hard enough that rank errors show) go through
  (s) the split-precision HIP encoder (MODEL.ENCODER_PRECISION split, the default and the mode bench.py times: fp16
      operand pairs, three products per multiply-add on the fp16 matrix cores) -> HIP normalise -> exact distance /
      re-ranking -> eval,
  (a) the all-fp32 HIP encoder (MODEL.ENCODER_PRECISION fp32: exact fp32 matrix instruction), same tail,
  (b) the fp16-MFMA HIP encoder (the throughput path), same tail,
  (o) the fp32 ORACLE pipeline on the host (torch CPU ViT restatement -> oracle normalise / distance / re-rank / eval).
Measured on MI355X (tools/map_parity.py): fp16 path |dmAP| = 1.3e-4 (features 4.3e-4 relative L2: the operand rounding
of 24 GEMMs), Rank-1 identical; fp32 path at the oracle's own rounding level.  So the bound of 1e-4 is asserted for the
fp32 mode, and the fp16 mode is held to the bound its feature error supports (5e-4), with the feature error itself
bounded -- stated here instead of hidden."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def data():
    from mpreid import synth
    x, pid = synth.identity_images(128, 16, 0.55)
    sd = synth.vit_state_dict(synth.VIT_B16, seed=7)
    torch.set_num_threads(min(torch.get_num_threads(), 32))
    f_or = np.concatenate([orc.vit_features(sd, synth.VIT_B16, x[s:s + 64]) for s in range(0, len(pid), 64)])
    return x, pid, sd, f_or


def _evaluate(ops, feats, pid, nq, rerank):
    fn = ops.l2_normalize(feats)
    d = ops.re_ranking(fn[:nq], fn[nq:], 50, 15, 0.3)[0] if rerank else ops.euclidean_distance(fn[:nq], fn[nq:])
    return orc.eval_func(d.cpu().numpy(), pid[:nq], pid[nq:])


@pytest.mark.parametrize("rerank", [False, True])
def test_image_to_map_parity(data, rerank):
    from mpreid import ops, synth
    x, pid, sd, f_or = data
    n = len(pid)
    nq = n // 5
    fo = orc.l2_normalize(f_or)
    d_or = orc.re_ranking(fo[:nq], fo[nq:], 50, 15, 0.3) if rerank else orc.euclidean_distance(fo[:nq], fo[nq:])
    cmc_o, map_o = orc.eval_func(d_or, pid[:nq], pid[nq:])
    assert 0.3 < map_o < (0.97 if rerank else 0.9), map_o   # hard enough to be informative (re-ranking lifts it)
    res = {}
    for prec in ("split", "fp32", "fp16"):
        enc = ops.VitEncoder(synth.VIT_B16, sd, (256, 128), precision=prec)
        f = torch.empty((n, enc.feat_dim), device="cuda")
        for s in range(0, n, 508):
            enc(torch.from_numpy(x[s:s + 508]), out=f[s:s + 508])
        rel = float(np.linalg.norm(f.cpu().numpy() - f_or) / np.linalg.norm(f_or))
        cmc, mAP = _evaluate(ops, f, pid, nq, rerank)
        res[prec] = (rel, abs(mAP - map_o), abs(float(cmc[0]) - float(cmc_o[0])), float(np.abs(cmc - cmc_o).max()))
        del enc
    print("image->mAP parity (rerank=%s): oracle mAP %.6f R1 %.6f | " % (rerank, map_o, cmc_o[0]) +
          " | ".join(f"{k}: feat rel-L2 {v[0]:.2e} dmAP {v[1]:.2e} dR1 {v[2]:.2e} max dCMC {v[3]:.2e}" for k, v in res.items()))
    for prec in ("split", "fp32"):   # north_star bound: the split mode (default, the one bench.py times) and the fp32 mode meet it
        rel, dmap, dr1, dcmc = res[prec]
        assert rel <= 2e-5 and dmap <= 1e-4 and dr1 <= 1e-4, (prec, res[prec])
    rel, dmap, dr1, dcmc = res["fp16"]
    assert rel <= 1e-3 and dmap <= 5e-4 and dr1 <= 1.0 / nq + 1e-9, res["fp16"]  # the bound the fp16 operands support


def test_fp32_encoder_small_config_and_options(golden):
    """the fp32 mode against the reference's own outputs (tests/golden/vit.npz): reduced config, camera embedding,
    stride 12 (L = 211), at fp32 accuracy"""
    from mpreid import ops, synth
    g = golden("vit.npz")
    small = dict(h_res=4, w_res=2, patch=16, stride=16, width=128, layers=2, heads=2, out_dim=64)
    enc = ops.VitEncoder(small, synth.vit_state_dict(small, seed=7, std=0.05, ln_jitter=0.1), (64, 32), precision="fp32")
    f = enc(torch.from_numpy(synth.synthetic_images(3, 64, 32, seed=3))).cpu().numpy()
    assert np.abs(f - g["small_feat"]).max() <= 2e-5
    big = synth.VIT_B16
    sd = synth.vit_state_dict(big, seed=7, std=0.02, ln_jitter=0.05)
    imgs = torch.from_numpy(synth.synthetic_images(4, 256, 128, seed=1234))
    enc = ops.VitEncoder(big, sd, (256, 128), precision="fp32")
    assert np.abs(enc(imgs).cpu().numpy() - g["b16_feat"]).max() <= 5e-5
    assert np.abs(enc(imgs, cv_emb=torch.from_numpy(g["b16_cv"])).cpu().numpy() - g["b16_feat_cv"]).max() <= 5e-5
    s12 = dict(big, h_res=21, w_res=10, stride=12)
    enc = ops.VitEncoder(s12, synth.vit_state_dict(s12, seed=8, std=0.02, ln_jitter=0.05), (256, 128), precision="fp32")
    assert np.abs(enc(imgs[:2]).cpu().numpy() - g["b16_s12_feat"]).max() <= 5e-5
