"""GPU parity of the RN50 (CLIP ModifiedResNet) path: single folded conv layers against torch fp32 conv + BatchNorm,
then the whole encoder against the reference's golden features and the fp32 oracle.  Floating-point kernels
(fp16 MFMA operands and fp16 activations, fp32 accumulation): tolerances are stated per test."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _rel(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return np.linalg.norm(got - want) / np.linalg.norm(want)


@pytest.mark.parametrize("B,H,W,cin,cout,k,with_id,relu", [
    (2, 16, 8, 64, 64, 3, False, True),      # layer1 conv2 shape, cout < 128 (masked columns)
    (3, 8, 4, 128, 128, 3, False, True),
    (2, 16, 8, 64, 256, 1, True, True),      # conv3 + residual + relu
    (1, 5, 7, 192, 72, 3, True, False),      # odd spatial size: M = 35 < one tile, borders everywhere
    (2, 32, 16, 256, 64, 1, False, True),
    (5, 16, 8, 512, 512, 3, False, True),    # several K steps per tap, several N tiles
])
def test_conv_layer_vs_torch_fp32(B, H, W, cin, cout, k, with_id, relu):
    """one conv + folded BatchNorm (+ identity) (+ ReLU): the inputs are rounded to fp16 first, so the only
    differences from the fp32 reference are the fp16 rounding of the folded weights and of the output:
    relative L2 <= 1e-3, max |d| <= 2e-3 * max|ref| + 2e-3"""
    from mpreid import ops
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + cin + cout + k)
    x = torch.randn((B, cin, H, W), generator=g).half().float()
    w = torch.randn((cout, cin, k, k), generator=g) * (2.0 / (cin * k * k)) ** 0.5
    bn = (1 + 0.1 * torch.randn(cout, generator=g), 0.1 * torch.randn(cout, generator=g),
          0.1 * torch.randn(cout, generator=g), 0.5 + torch.rand(cout, generator=g))
    idt = torch.randn((B, cout, H, W), generator=g).half().float() if with_id else None
    ref = F.batch_norm(F.conv2d(x, w, None, padding=k // 2), bn[2], bn[3], bn[0], bn[1], training=False, eps=1e-5)
    if with_id:
        ref = ref + idt
    if relu:
        ref = F.relu(ref)
    wk, bk = ops.fold_conv_bn(w.numpy(), tuple(t.numpy() for t in bn))
    act = x.permute(0, 2, 3, 1).contiguous().half().cuda()
    idn = None if idt is None else idt.permute(0, 2, 3, 1).contiguous().half().cuda()
    out = ops.conv_f16_nhwc(act, wk.cuda(), bk.cuda(), cout, k * k, identity=idn, relu=relu)
    got = out.float().cpu().permute(0, 3, 1, 2).numpy()
    assert _rel(got, ref.numpy()) <= 1e-3, _rel(got, ref.numpy())
    assert np.abs(got - ref.numpy()).max() <= 2e-3 * float(ref.abs().max()) + 2e-3


SMALL = dict(layers=(1, 2, 1, 1), width=16, heads=8, out_dim=64, h_res=4, w_res=2)


def _check(got, want, rel):
    r = _rel(got, want)
    cos = (got * want).sum(1) / (np.linalg.norm(got, axis=1) * np.linalg.norm(want, axis=1))
    assert r <= rel and cos.min() >= 0.9999, (r, cos.min())
    return r


def test_rn50_small_vs_reference(golden):
    """reduced ModifiedResNet (every channel count padded to 64 in storage) against the reference class's output:
    fp16 activations + fp16 MFMA operands through ~20 layers: relative L2 <= 5e-3"""
    from mpreid import ops, synth
    g = golden("rn50.npz")
    enc = ops.Rn50Encoder(SMALL, synth.rn50_state_dict(SMALL, seed=11), (64, 32), precision="fp16")
    f = enc(torch.from_numpy(synth.synthetic_images(3, 64, 32, seed=31))).cpu().numpy()
    assert f.shape == (3, 576)
    print("rn50 small rel-L2:", _check(f, g["small_feat"], 5e-3))


def test_rn50_full_vs_reference(golden):
    from mpreid import ops, synth
    g = golden("rn50.npz")
    sd = synth.rn50_state_dict(synth.RN50, seed=11)
    enc = ops.Rn50Encoder(synth.RN50, sd, (256, 128), precision="fp16")
    imgs = synth.synthetic_images(3, 256, 128, seed=32)
    f = enc(torch.from_numpy(imgs)).cpu().numpy()
    assert f.shape == (3, 3072)
    print("rn50 rel-L2 vs reference:", _check(f, g["rn50_feat"], 5e-3))
    # batch independence: the same images inside a larger batch give the same rows bit for bit
    more = synth.synthetic_images(9, 256, 128, seed=99)
    more[4:7] = imgs
    f2 = enc(torch.from_numpy(more)).cpu().numpy()
    assert np.array_equal(f2[4:7], f)


def test_rn50_u8_and_bn_neck_vs_oracle():
    from mpreid import ops, synth
    sd = synth.rn50_state_dict(SMALL, seed=12)
    rng = np.random.default_rng(4)
    bn = {n: (1 + 0.1 * rng.standard_normal(d).astype(np.float32), 0.1 * rng.standard_normal(d).astype(np.float32),
              0.1 * rng.standard_normal(d).astype(np.float32), (0.5 + rng.random(d)).astype(np.float32))
          for n, d in (("bottleneck", 512), ("bottleneck_proj", 64))}
    enc = ops.Rn50Encoder(SMALL, sd, (64, 32), neck_after=True, bn=bn, precision="fp16")
    u8 = rng.integers(0, 256, (5, 64, 32, 3), dtype=np.uint8)
    mean, std = (0.5, 0.4, 0.45), (0.5, 0.25, 0.3)
    t = torch.from_numpy(u8).permute(0, 3, 1, 2).float().div(255)
    t = (t - torch.tensor(mean)[None, :, None, None]) / torch.tensor(std)[None, :, None, None]
    want = orc.rn50_features(sd, SMALL, t.numpy(), bn=bn, neck_feat="after")
    got_u8 = enc.forward_u8(torch.from_numpy(u8), mean, std).cpu().numpy()
    got_f32 = enc(t.contiguous()).cpu().numpy()
    assert np.array_equal(got_u8, got_f32)       # ToTensor + Normalize fused into the first conv: same bits
    _check(got_u8, want, 5e-3)


def test_rn50_other_resolution_vs_oracle():
    """a non-square grid that is not a power of two (96x64 input -> 6x4 = 24 attention-pool tokens, M not a
    multiple of the 128-row GEMM tile in the deeper layers), ragged batch of 5"""
    from mpreid import ops, synth
    cfg = dict(layers=(1, 1, 2, 1), width=16, heads=8, out_dim=64, h_res=6, w_res=4)
    sd = synth.rn50_state_dict(cfg, seed=21)
    enc = ops.Rn50Encoder(cfg, sd, (96, 64), precision="fp16")
    imgs = synth.synthetic_images(5, 96, 64, seed=41)
    got = enc(torch.from_numpy(imgs)).cpu().numpy()
    want = orc.rn50_features(sd, cfg, imgs)
    _check(got, want, 5e-3)


# ---- fp32 mode (MODEL.ENCODER_PRECISION fp32): the parity mode of the RN50 tower ----
def test_rn50_fp32_mode_vs_reference(golden):
    """every activation fp32, every convolution a GEMM on the exact fp32 matrix instruction: the reference class's own
    outputs (tests/golden/rn50.npz) at fp32 accuracy -- relative L2 <= 2e-5 (the fp16 tower: 2.6e-3), reduced and full
    config, and rows independent of the batch they sit in"""
    from mpreid import ops, synth
    g = golden("rn50.npz")
    enc = ops.Rn50Encoder(SMALL, synth.rn50_state_dict(SMALL, seed=11), (64, 32), precision="fp32")
    f = enc(torch.from_numpy(synth.synthetic_images(3, 64, 32, seed=31))).cpu().numpy()
    assert f.shape == (3, 576)
    print("rn50 fp32 mode, small rel-L2:", _check(f, g["small_feat"], 2e-5))
    enc = ops.Rn50Encoder(synth.RN50, synth.rn50_state_dict(synth.RN50, seed=11), (256, 128), precision="fp32")
    imgs = synth.synthetic_images(3, 256, 128, seed=32)
    f = enc(torch.from_numpy(imgs)).cpu().numpy()
    assert f.shape == (3, 3072)
    print("rn50 fp32 mode, full rel-L2:", _check(f, g["rn50_feat"], 2e-5))
    more = synth.synthetic_images(70, 256, 128, seed=99)    # two workspace chunks (64 + 6)
    more[62:65] = imgs
    assert np.array_equal(enc(torch.from_numpy(more)).cpu().numpy()[62:65], f)


def test_rn50_fp32_mode_options_vs_oracle():
    """BN necks ('after'), uint8 input, a 6x4 grid with a ragged batch"""
    from mpreid import ops, synth
    sd = synth.rn50_state_dict(SMALL, seed=12)
    rng = np.random.default_rng(4)
    bn = {n: (1 + 0.1 * rng.standard_normal(d).astype(np.float32), 0.1 * rng.standard_normal(d).astype(np.float32),
              0.1 * rng.standard_normal(d).astype(np.float32), (0.5 + rng.random(d)).astype(np.float32))
          for n, d in (("bottleneck", 512), ("bottleneck_proj", 64))}
    enc = ops.Rn50Encoder(SMALL, sd, (64, 32), neck_after=True, bn=bn, precision="fp32")
    u8 = rng.integers(0, 256, (5, 64, 32, 3), dtype=np.uint8)
    mean, std = (0.5, 0.4, 0.45), (0.5, 0.25, 0.3)
    t = torch.from_numpy(u8).permute(0, 3, 1, 2).float().div(255)
    t = (t - torch.tensor(mean)[None, :, None, None]) / torch.tensor(std)[None, :, None, None]
    want = orc.rn50_features(sd, SMALL, t.numpy(), bn=bn, neck_feat="after")
    got_u8 = enc.forward_u8(torch.from_numpy(u8), mean, std).cpu().numpy()
    _check(got_u8, want, 2e-5)
    got_f = enc(t.contiguous()).cpu().numpy()
    _check(got_f, want, 2e-5)
    assert np.array_equal(got_u8, got_f)   # ToTensor + Normalize inside the stem's first convolution (round 5): same bits
    cfg = dict(layers=(1, 1, 2, 1), width=16, heads=8, out_dim=64, h_res=6, w_res=4)
    sd = synth.rn50_state_dict(cfg, seed=21)
    imgs = synth.synthetic_images(5, 96, 64, seed=41)
    got = ops.Rn50Encoder(cfg, sd, (96, 64), precision="fp32")(torch.from_numpy(imgs)).cpu().numpy()
    _check(got, orc.rn50_features(sd, cfg, imgs), 2e-5)


# ---- split mode (MODEL.ENCODER_PRECISION split, the default): fp32 activations, layer1-4 over fp16 pairs on the matrix cores ----
def test_rn50_split_mode_vs_reference(golden):
    """the reference class's own outputs (tests/golden/rn50.npz) at fp32 accuracy -- relative L2 <= 2e-5 -- from the fp16
    matrix cores: reduced config (every channel count padded: 8 .. 256 channels against 64-wide k segments and 128-wide
    tiles) and the full RN50; rows independent of the batch they sit in (two workspace chunks, ragged second chunk)"""
    from mpreid import ops, synth
    g = golden("rn50.npz")
    enc = ops.Rn50Encoder(SMALL, synth.rn50_state_dict(SMALL, seed=11), (64, 32), precision="split")
    f = enc(torch.from_numpy(synth.synthetic_images(3, 64, 32, seed=31))).cpu().numpy()
    assert f.shape == (3, 576)
    print("rn50 split mode, small rel-L2:", _check(f, g["small_feat"], 2e-5))
    enc = ops.Rn50Encoder(synth.RN50, synth.rn50_state_dict(synth.RN50, seed=11), (256, 128), precision="split")
    imgs = synth.synthetic_images(3, 256, 128, seed=32)
    f = enc(torch.from_numpy(imgs)).cpu().numpy()
    assert f.shape == (3, 3072)
    print("rn50 split mode, full rel-L2:", _check(f, g["rn50_feat"], 2e-5))
    more = synth.synthetic_images(261, 256, 128, seed=99)    # two workspace chunks (256 + 5)
    more[254:257] = imgs
    out = enc(torch.from_numpy(more)).cpu().numpy()
    assert np.isfinite(out).all() and np.array_equal(out[254:257], f)
    assert np.array_equal(enc(torch.from_numpy(more)).cpu().numpy(), out)   # run to run


def test_rn50_split_mode_options_vs_oracle():
    """BN necks ('after'), uint8 input, a 6x4 grid with a ragged batch, through make_model's default precision"""
    from mpreid import ops, synth
    sd = synth.rn50_state_dict(SMALL, seed=12)
    rng = np.random.default_rng(4)
    bn = {n: (1 + 0.1 * rng.standard_normal(d).astype(np.float32), 0.1 * rng.standard_normal(d).astype(np.float32),
              0.1 * rng.standard_normal(d).astype(np.float32), (0.5 + rng.random(d)).astype(np.float32))
          for n, d in (("bottleneck", 512), ("bottleneck_proj", 64))}
    enc = ops.Rn50Encoder(SMALL, sd, (64, 32), neck_after=True, bn=bn, precision="split")
    u8 = rng.integers(0, 256, (5, 64, 32, 3), dtype=np.uint8)
    mean, std = (0.5, 0.4, 0.45), (0.5, 0.25, 0.3)
    t = torch.from_numpy(u8).permute(0, 3, 1, 2).float().div(255)
    t = (t - torch.tensor(mean)[None, :, None, None]) / torch.tensor(std)[None, :, None, None]
    want = orc.rn50_features(sd, SMALL, t.numpy(), bn=bn, neck_feat="after")
    got_u8 = enc.forward_u8(torch.from_numpy(u8), mean, std).cpu().numpy()
    _check(got_u8, want, 2e-5)
    got_f = enc(t.contiguous()).cpu().numpy()
    _check(got_f, want, 2e-5)
    assert np.array_equal(got_u8, got_f)   # ToTensor + Normalize inside the stem's first convolution (round 5): same bits
    cfg = dict(layers=(1, 1, 2, 1), width=16, heads=8, out_dim=64, h_res=6, w_res=4)
    sd = synth.rn50_state_dict(cfg, seed=21)
    imgs = synth.synthetic_images(5, 96, 64, seed=41)
    got = ops.Rn50Encoder(cfg, sd, (96, 64), precision="split")(torch.from_numpy(imgs)).cpu().numpy()
    _check(got, orc.rn50_features(sd, cfg, imgs), 2e-5)


def test_rn50_image_to_map_parity():
    """image -> mAP for MODEL.NAME RN50 (north_star: within 1e-4 of the reference CPU path): identity-structured synthetic
    images through the fp32-mode tower -> normalise -> distance / re-ranking -> eval against the fp32 oracle pipeline.
    A random-init post-ReLU tower maps every image close to one direction (median normalised distance 0.007: the
    DEGENERATE geometry of tests/test_gpu_map_parity.py, where a 1e-6 feature error is a 1e-4 relative distance error
    and the reference does not agree with itself to 1e-4: the same fp32 oracle gives a re-ranked mAP of 0.6366 on this
    container's CPU and 0.6349 on the GPU box's).  So the FEATURES are held to the fp32 level (2e-5; measured 4.1e-6) and
    the metrics to 5e-4 / one query (measured on MI355X: |dmAP| 8.9e-5 Euclidean, 8.0e-5 re-ranked, no query differs);
    the fp16 tower is reported and held to what its 2.3e-3 feature error supports."""
    from mpreid import ops, synth
    n_ids, per_id = 64, 8
    x, pid = synth.identity_images(n_ids, per_id, 0.2)   # oracle: Euclidean mAP 0.59 / Rank-1 0.76, re-ranked 0.64 / 0.75
    sd = synth.rn50_state_dict(synth.RN50, seed=11)
    torch.set_num_threads(min(torch.get_num_threads(), 32))
    f_or = np.concatenate([orc.rn50_features(sd, synth.RN50, x[s:s + 32]) for s in range(0, len(pid), 32)])
    n = len(pid)
    nq = n // 4
    fo = orc.l2_normalize(f_or)
    res = {}
    for prec in ("split", "fp32", "fp16"):
        enc = ops.Rn50Encoder(synth.RN50, sd, (256, 128), precision=prec)
        f = enc(torch.from_numpy(x))
        res[prec] = [float(np.linalg.norm(f.cpu().numpy() - f_or) / np.linalg.norm(f_or))]
        fn = ops.l2_normalize(f)
        for rerank in (False, True):
            d_or = orc.re_ranking(fo[:nq], fo[nq:], 20, 6, 0.3) if rerank else orc.euclidean_distance(fo[:nq], fo[nq:])
            cmc_o, map_o = orc.eval_func(d_or, pid[:nq], pid[nq:])
            d = ops.re_ranking(fn[:nq], fn[nq:], 20, 6, 0.3)[0] if rerank else ops.euclidean_distance(fn[:nq], fn[nq:])
            cmc, mAP = orc.eval_func(d.cpu().numpy(), pid[:nq], pid[nq:])
            res[prec] += [map_o, abs(mAP - map_o), abs(float(cmc[0]) - float(cmc_o[0]))]
        del enc
    print("rn50 image->mAP: median distance %.4f | " % float(np.median(orc.euclidean_distance(fo[:nq], fo[nq:]))) +
          " | ".join(f"{k}: feat rel-L2 {v[0]:.2e}; euclid mAP {v[1]:.4f} dmAP {v[2]:.2e} dR1 {v[3]:.2e}; "
                     f"rerank mAP {v[4]:.4f} dmAP {v[5]:.2e} dR1 {v[6]:.2e}" for k, v in res.items()))
    # features at the fp32 level, and metrics that move no more than a RANDOM feature error of that size moves them on this
    # (degenerate: median normalised distance 0.007) set -- conftest.map_noise_envelope: the largest |dmAP| / |dRank-1| of four
    # random perturbations of the oracle's features at the modes' measured error size; bound = 1e-4 + 3x that: no systematic bias
    from conftest import map_noise_envelope
    rel = max(res["split"][0], res["fp32"][0])
    assert rel <= 2e-5, res
    env_e = map_noise_envelope(orc, f_or, rel, pid, nq, False, 20, 6, seeds=4)
    env_r = map_noise_envelope(orc, f_or, rel, pid, nq, True, 20, 6, seeds=4)
    print("rn50: noise envelope of a %.1e feature error: euclid dmAP %.2e dR1 %.2e; re-ranked dmAP %.2e dR1 %.2e" %
          (rel, env_e[0], env_e[1], env_r[0], env_r[1]))
    # (mAP over 128 queries moves in quanta -- one near-tie flip in one query's ranking is ~1e-3 -- and the fp32 ORACLE itself
    # gives 0.6366 on the build container's CPU and 0.6349 on the GPU box's for the re-ranked mAP of this set: 1.7e-3 apart.
    # So nothing below 2e-3 can be asserted of ANY implementation here; the envelope term catches a set that got noisier.)
    for prec in ("split", "fp32"):
        r = res[prec]
        assert r[2] <= max(2e-3, 1e-4 + 3.0 * env_e[0]) and r[5] <= max(2e-3, 1e-4 + 3.0 * env_r[0]), (prec, r, env_e, env_r)
        assert r[3] <= 1.0 / nq + 3.0 * env_e[1] + 1e-9 and r[6] <= 1.0 / nq + 3.0 * env_r[1] + 1e-9, (prec, r, env_e, env_r)
    r = res["fp16"]
    assert r[0] <= 5e-3 and max(r[2], r[5]) <= 2e-2, r


@pytest.mark.parametrize("precision", ["split", "fp32"])
def test_rn50_views_inside_the_stem_match_materialised_tensors(precision):
    """the test-time-augmentation views (processor/processor_uniprompt_stage2.py:605-633) inside the stem's first convolution
    (mpreid_rn50_forward_split_view / _f32_view, round 5) == the view tensors the reference materialises (torch ops) through the
    plain entry point, bit for bit -- fp32 and uint8 input, all three views; and through make_model's _encode"""
    from mpreid import ops, synth
    rng = np.random.default_rng(3)
    sd = synth.rn50_state_dict(SMALL, seed=12)
    enc = ops.Rn50Encoder(SMALL, sd, (64, 32), precision=precision)
    u8 = rng.integers(0, 256, (5, 64, 32, 3), dtype=np.uint8)
    mean, std = (0.5, 0.4, 0.45), (0.5, 0.25, 0.3)
    x = torch.from_numpy(u8).permute(0, 3, 1, 2).float().div(255)
    x = ((x - torch.tensor(mean)[None, :, None, None]) / torch.tensor(std)[None, :, None, None]).contiguous()
    views = {ops.VIEW_FLIP: torch.flip(x, [3]).contiguous(),
             ops.VIEW_PSEUDO_IR: x.mean(dim=1, keepdim=True).repeat(1, 3, 1, 1).contiguous(),
             ops.VIEW_PSEUDO_RGB: x[:, 0:1].repeat(1, 3, 1, 1).contiguous()}
    for v, t in views.items():
        want = enc(t).cpu().numpy()
        assert np.array_equal(enc.forward_view(x, v).cpu().numpy(), want), v
        assert np.array_equal(enc.forward_view(torch.from_numpy(u8), v, None, mean, std).cpu().numpy(), want), v


def test_rn50_image_to_map_parity_spread_set():
    """north_star's plain 1e-4 on mAP / Rank-1 for MODEL.NAME RN50, on a set where it MEANS something (round-4 advisor: the
    degenerate set above lets a systematic bias of 2e-3 pass).  The degenerate geometry of a random-init ResNet has two
    causes, both removed here the way training removes them: (1) synth.rn50_state_dict draws the BatchNorm running statistics
    at random -- oracle.rn50_calibrate_bn sets every layer's statistics to those of its own input over 64 of the images (what a
    trained network's BatchNorm layers hold); (2) all 2048 pooled post-ReLU channels are positive for every image -- the last
    block's channels are made selective (its bn3 bias lowered so that a channel's pre-activation sits one standard deviation
    below zero on average) and the projected part is centred.  Low-frequency identity templates (8 x 4 colour grids: pooled
    convolutional features cannot tell white-noise templates apart).  Result: 1024 images, normalised distances 0.1 ... 1.2,
    median ~0.3 -- the spread of a trained re-id model -- Euclidean mAP 0.60, re-ranked 0.69.

    Asserted for the split (default) and the fp32 tower: features <= 2e-5 relative L2; WITHOUT re-ranking |dmAP| <= 1e-4 and
    |dRank-1| <= 1e-4 against the fp32 oracle (measured <= 6e-7, no query).  WITH re-ranking (k1 20, k2 6) the fp32 ORACLE is
    its own noise source on this set: the same graph evaluated in float64 (features 4.3e-6 away, oracle.rn50_features(dtype=
    'float64')) moves the re-ranked mAP by 3.5e-4 -- one k-reciprocal membership flipping -- so the 1e-4 is asserted against
    that exact-arithmetic answer (measured: split 3.8e-5, fp32 tower 8.4e-5, no query's Rank-1; reproducible, the float64 graph
    does not depend on the host's BLAS path) and the distance to the fp32 oracle is held to 1e-4 + the oracle's OWN measured
    distance from the float64 answer (triangle inequality; measured 3.9e-4 / 2.6e-4 against 1e-4 + 3.5e-4; the noise envelope of
    a random feature error of the modes' size, 2.8e-4, is printed beside it).  Reference: model/clip/model.py:10-148, model/make_model.py:82-86, utils/metrics.py:28-88."""
    from conftest import map_noise_envelope
    from mpreid import ops, synth
    n_ids, per_id = 128, 8
    x, pid = synth.identity_images(n_ids, per_id, 0.5, grid=(8, 4))
    torch.set_num_threads(min(torch.get_num_threads(), 32))
    sd = orc.rn50_calibrate_bn(synth.rn50_state_dict(synth.RN50, seed=11), synth.RN50, x[:64], selective=1.0)
    n = len(pid)
    nq = n // 4
    f_or = np.concatenate([orc.rn50_features(sd, synth.RN50, x[s:s + 32]) for s in range(0, n, 32)])
    f_64 = np.concatenate([orc.rn50_features(sd, synth.RN50, x[s:s + 32], dtype="float64") for s in range(0, n, 32)])
    fo, fo64 = orc.l2_normalize(f_or), orc.l2_normalize(f_64.astype(np.float32))
    d_plain = orc.euclidean_distance(fo[:nq], fo[nq:])
    med = float(np.median(d_plain))
    assert 0.1 < med < 1.0, med          # the spread geometry (the degenerate set: 0.007)
    ref = {False: orc.eval_func(d_plain, pid[:nq], pid[nq:]),
           True: orc.eval_func(orc.re_ranking(fo[:nq], fo[nq:], 20, 6, 0.3), pid[:nq], pid[nq:]),
           "rr64": orc.eval_func(orc.re_ranking(fo64[:nq], fo64[nq:], 20, 6, 0.3), pid[:nq], pid[nq:])}
    assert 0.15 < ref[False][1] < 0.97 and 0.15 < ref[True][1] < 0.97
    res = {}
    for prec in ("split", "fp32", "fp16"):
        enc = ops.Rn50Encoder(synth.RN50, sd, (256, 128), precision=prec)
        f = torch.cat([enc(torch.from_numpy(x[s:s + 256])) for s in range(0, n, 256)])
        rel = float(np.linalg.norm(f.cpu().numpy() - f_or) / np.linalg.norm(f_or))
        fn = ops.l2_normalize(f)
        cmc_e, map_e = orc.eval_func(ops.euclidean_distance(fn[:nq], fn[nq:]).cpu().numpy(), pid[:nq], pid[nq:])
        cmc_r, map_r = orc.eval_func(ops.re_ranking(fn[:nq], fn[nq:], 20, 6, 0.3)[0].cpu().numpy(), pid[:nq], pid[nq:])
        res[prec] = dict(rel=rel, e_dmap=abs(map_e - ref[False][1]), e_dr1=abs(float(cmc_e[0]) - float(ref[False][0][0])),
                         r_dmap=abs(map_r - ref[True][1]), r_dr1=abs(float(cmc_r[0]) - float(ref[True][0][0])),
                         r64_dmap=abs(map_r - ref["rr64"][1]), r64_dr1=abs(float(cmc_r[0]) - float(ref["rr64"][0][0])))
        del enc
    rel_max = max(res["split"]["rel"], res["fp32"]["rel"])
    env = map_noise_envelope(orc, f_or, rel_max, pid, nq, True, 20, 6, seeds=4)
    bound_rr = min(5e-4, 1e-4 + 2.0 * env[0])
    print("rn50 image->mAP [spread, %d images]: median distance %.4f, oracle mAP %.4f / re-ranked %.4f (float64 graph %.4f: %.1e away), "
          "re-ranked noise envelope %.2e -> bound %.2e | " % (n, med, ref[False][1], ref[True][1], ref["rr64"][1],
                                                             abs(ref[True][1] - ref["rr64"][1]), env[0], bound_rr) +
          " | ".join(f"{k}: " + " ".join(f"{a} {b:.2e}" for a, b in v.items()) for k, v in res.items()))
    for prec in ("split", "fp32"):
        r = res[prec]
        assert r["rel"] <= 2e-5, (prec, r)
        assert r["e_dmap"] <= 1e-4 and r["e_dr1"] <= 1e-4, (prec, r)                  # north_star, plain
        assert r["r64_dmap"] <= 1e-4 and r["r64_dr1"] <= 1e-4, (prec, r)             # re-ranked: against exact arithmetic
        # vs the fp32 oracle: by the triangle inequality no further than the oracle's own distance from exact arithmetic (host
        # dependent: its BLAS path) + the 1e-4 asserted above; the capped envelope bound of round 5's first version is printed
        oracle_noise = abs(ref[True][1] - ref["rr64"][1])
        assert r["r_dmap"] <= 1e-4 + oracle_noise + 1e-12 and r["r_dr1"] <= 1.0 / nq + 2.0 * env[1] + 1e-9, (prec, r, env, oracle_noise)
    r = res["fp16"]
    assert r["rel"] <= 1e-2 and max(r["e_dmap"], r["r_dmap"]) <= 5e-2, r     # (reported: what fp16 activations support)
