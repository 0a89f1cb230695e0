"""Image -> mAP parity (north_star: "matching the reference CPU path's mAP / Rank-1 within 1e-4").

Synthetic images with identity structure (mpreid.synth.identity_images) go through
  (s) the split-precision HIP encoder (MODEL.ENCODER_PRECISION split, the default and the mode bench.py times: fp16
      operand pairs, three products per multiply-add on the fp16 matrix cores) -> HIP normalise -> exact distance /
      re-ranking -> eval,
  (a) the all-fp32 HIP encoder (MODEL.ENCODER_PRECISION fp32: exact fp32 matrix instruction), same tail,
  (b) the fp16-MFMA HIP encoder (single fp16 operands: the fastest mode), same tail,
  (o) the fp32 ORACLE pipeline on the host (torch CPU ViT restatement -> oracle normalise / distance / re-rank / eval).

Two sets of weights, because what "within 1e-4" can mean depends on the geometry of the features:

SPREAD (the parity bar).  Weights ~ N(0, 0.05^2): the image content drives the features, normalised distances are
  0.15 ... 0.6 with a median of 0.35 -- the spread a trained re-id model has -- and with beta = 0.4 the task is hard
  (1024 images: Euclidean mAP 0.28, with re-ranking 0.40, Rank-1 0.60).  Here the reference's OWN fp32 rounding is invisible in the metrics: the oracle
  evaluated in fp64 instead of fp32 (features 1.07e-6 apart) moves mAP by 2.7e-5 and no Rank-1 (tools/map_noise_floor.py
  spread).  So 1e-4 is a meaningful bound, and it is ASSERTED for the split
  and the fp32 mode, with and without re-ranking: |dmAP| <= 1e-4, |dRank-1| <= 1e-4 (i.e. not one query differs),
  features within 2e-5 relative L2 (measured on MI355X: split 1.5e-6 ... 3.7e-6 / |dmAP| 2.7e-8 ... 5.3e-5 / no query
  differs; fp32 2.1e-6 / 2.2e-8 ... 5.0e-5 -- the spread is the ORACLE's: the GPU boxes of the pool have different host
  CPUs and the fp32 torch graph takes different MKL paths on them; both HIP modes move together).  The fp16 mode is reported and held to the bound its feature error supports (measured 8.9e-4 / |dmAP| 5.4e-4 /
  one query: single fp16 operands do NOT meet 1e-4 on realistic geometry either).

DEGENERATE (the round-1/2 set, kept as a stress case).  Weights ~ N(0, 0.02^2): the CLS row barely sees the image, all
  features are nearly parallel (normalised distances 0.008 ... 0.03), and a feature error of 1e-6 is a distance error of
  ~1e-4 relative.  On this set the reference's own arithmetic noise exceeds 1e-4: oracle fp32 vs the same graph in fp64
  (features 6.1e-7 apart) differ by 1.7e-4 in mAP and by one query in Rank-1; the fp32 oracle on two different CPUs
  (this container / the GPU box: another MKL code path) by 1.3e-4 (tools/map_noise_floor.py).  No implementation can be
  held to 1e-4 against a reference that does not agree with itself to 1e-4, so here the FEATURES are held to the fp32
  level (2e-5; measured 1.4e-6 split, 1.3e-6 fp32) and the metrics to the set's own noise envelope (5e-4, one query)."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

SETS = {
    # name: (identities, images per identity, beta, weight std)
    "big": (512, 8, 0.4, 0.05),          # 4096 images, the spread geometry (test_image_to_map_parity_4096_images)
    "spread": (128, 8, 0.4, 0.05),       # = the 1024 images of identities 0 .. 127 of "big" (ONE oracle encode serves both)
    "degenerate": (128, 8, 0.55, 0.02),  # (round 5: 1024 images instead of 2048 -- the suite's time went to the 4096-image set;
                                         #  the non-parity fp16 mode's bound below follows the coarser quanta: one query = 4.9e-3)
}


@pytest.fixture(scope="module")
def data():
    from mpreid import synth
    torch.set_num_threads(min(torch.get_num_threads(), 32))
    cache = {}

    def get(name):
        if name not in cache:
            if name == "spread":
                x, pid, sd, f_or = get("big")
                keep = np.nonzero(pid < SETS["spread"][0])[0]
                cache[name] = (x[keep], pid[keep], sd, f_or[keep])
                return cache[name]
            n_ids, per_id, beta, std = SETS[name]
            x, pid = synth.identity_images(n_ids, per_id, beta)
            sd = synth.vit_state_dict(synth.VIT_B16, seed=7, std=std)
            f_or = np.concatenate([orc.vit_features(sd, synth.VIT_B16, x[s:s + 64]) for s in range(0, len(pid), 64)])
            cache[name] = (x, pid, sd, f_or)
        return cache[name]
    return get


def _evaluate(ops, feats, pid, nq, rerank):
    fn = ops.l2_normalize(feats)
    d = ops.re_ranking(fn[:nq], fn[nq:], 50, 15, 0.3)[0] if rerank else ops.euclidean_distance(fn[:nq], fn[nq:])
    return orc.eval_func(d.cpu().numpy(), pid[:nq], pid[nq:])


@pytest.mark.parametrize("rerank", [False, True])
@pytest.mark.parametrize("name", ["spread", "degenerate"])
def test_image_to_map_parity(data, name, rerank):
    from mpreid import ops, synth
    x, pid, sd, f_or = data(name)
    n = len(pid)
    nq = n // 5
    fo = orc.l2_normalize(f_or)
    d_or = orc.re_ranking(fo[:nq], fo[nq:], 50, 15, 0.3) if rerank else orc.euclidean_distance(fo[:nq], fo[nq:])
    cmc_o, map_o = orc.eval_func(d_or, pid[:nq], pid[nq:])
    assert 0.2 < map_o < 0.97, map_o   # hard enough to be informative (re-ranking lifts it)
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
    print("image->mAP parity [%s] (rerank=%s): oracle mAP %.6f R1 %.6f median distance %.4f | " %
          (name, rerank, map_o, cmc_o[0], float(np.median(orc.euclidean_distance(fo[:nq], fo[nq:])))) +
          " | ".join(f"{k}: feat rel-L2 {v[0]:.2e} dmAP {v[1]:.2e} dR1 {v[2]:.2e} max dCMC {v[3]:.2e}" for k, v in res.items()))
    if name == "spread":
        for prec in ("split", "fp32"):   # north_star's bound: the default (measured) mode and the fp32 mode meet it
            rel, dmap, dr1, dcmc = res[prec]
            assert rel <= 2e-5 and dmap <= 1e-4 and dr1 <= 1e-4, (prec, res[prec])
        rel, dmap, dr1, dcmc = res["fp16"]
        assert rel <= 3e-3 and dmap <= 3e-3 and dr1 <= 2.0 / nq + 1e-9, res["fp16"]   # what single fp16 operands support
    else:
        # the degenerate set: the bound is DERIVED -- the noise envelope of a random feature error of the size the mode
        # actually has (conftest.map_noise_envelope: 2x the largest |dmAP| of four realisations + north_star's 1e-4), never above
        # the 5e-4 envelope rounds 1-3 asserted
        from conftest import map_noise_envelope
        rel_max = max(res["split"][0], res["fp32"][0])
        env = map_noise_envelope(orc, f_or, rel_max, pid, nq, rerank, 50, 15, seeds=4)
        bound = min(5e-4, 1e-4 + 2.0 * env[0])
        print("noise envelope [%s] (rerank=%s): a %.1e feature error moves mAP by up to %.2e, Rank-1 by %.2e -> bound %.2e" %
              (name, rerank, rel_max, env[0], env[1], bound))
        for prec in ("split", "fp32"):
            rel, dmap, dr1, dcmc = res[prec]
            assert rel <= 2e-5 and dmap <= bound and dr1 <= 1.0 / nq + 2.0 * env[1] + 1e-9, (prec, res[prec], env, bound)
        rel, dmap, dr1, dcmc = res["fp16"]
        assert rel <= 1e-3 and dmap <= 3e-3 and dr1 <= 2.0 / nq + 1e-9, res["fp16"]   # (measured 4.3e-4 / 1.4e-3 / one query)


def test_image_to_map_parity_4096_images(data):
    """the spread geometry at four times the size (512 ids x 8 = 4096 images, 819 queries; round 4 ran this once as
    tools/map_parity_large.py, round 5 asserts it): split (the default, measured mode) and fp32 within north_star's 1e-4 on
    mAP and Rank-1 -- not one of 819 queries -- with and without re-ranking (k1 50, k2 15, lambda 0.3), features within 2e-5.
    The oracle encode of 4096 images takes ~3 minutes on the GPU box's host cores: the suite's longest test."""
    from mpreid import ops, synth
    x, pid, sd, f_or = data("big")
    n = len(pid)
    nq = n // 5
    fo = orc.l2_normalize(f_or)
    ref = {}
    for rr in (False, True):
        d = orc.re_ranking(fo[:nq], fo[nq:], 50, 15, 0.3) if rr else orc.euclidean_distance(fo[:nq], fo[nq:])
        ref[rr] = orc.eval_func(d, pid[:nq], pid[nq:])
        assert 0.1 < ref[rr][1] < 0.9
    out = []
    for prec in ("split", "fp32", "fp16"):
        enc = ops.VitEncoder(synth.VIT_B16, sd, (256, 128), precision=prec)
        f = torch.empty((n, enc.feat_dim), device="cuda")
        for s in range(0, n, 508):
            enc(torch.from_numpy(x[s:s + 508]), out=f[s:s + 508])
        rel = float(np.linalg.norm(f.cpu().numpy() - f_or) / np.linalg.norm(f_or))
        for rr in (False, True):
            cmc, mAP = _evaluate(ops, f, pid, nq, rr)
            dmap, dr1 = abs(mAP - ref[rr][1]), abs(float(cmc[0]) - float(ref[rr][0][0]))
            out.append(f"{prec} rerank={rr}: rel-L2 {rel:.2e} |dmAP| {dmap:.2e} |dR1| {dr1:.2e} ({round(dr1 * nq)} of {nq})")
            if prec != "fp16":
                assert rel <= 2e-5 and dmap <= 1e-4 and dr1 <= 1e-4, (prec, rr, rel, dmap, dr1)
        del enc
    print("image->mAP parity, 4096 images: oracle mAP %.4f / re-ranked %.4f | " % (ref[False][1], ref[True][1]) + " | ".join(out))


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
