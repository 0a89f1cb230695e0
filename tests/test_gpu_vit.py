"""GPU parity of the HIP ViT encoder (C ABI) against the reference's golden features and the fp32
oracle.  The encoder computes its GEMMs with fp16 operands / fp32 accumulation (residual stream,
LayerNorm statistics and softmax in fp32), so this is a floating-point kernel with a stated
tolerance: relative L2 error <= 4e-3 and max |d| <= 3e-2 on O(1) features, cosine >= 0.99999."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

SMALL = dict(h_res=4, w_res=2, patch=16, stride=16, width=128, layers=2, heads=2, out_dim=64)


def _close(got, want, rel=4e-3, mx=3e-2):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    rl2 = np.linalg.norm(got - want) / np.linalg.norm(want)
    cos = (got * want).sum(1) / (np.linalg.norm(got, axis=1) * np.linalg.norm(want, axis=1))
    assert rl2 <= rel and np.abs(got - want).max() <= mx and cos.min() >= 0.99999, (rl2, np.abs(got - want).max(),
                                                                                   cos.min())
    return rl2


def _encoder(cfg, sd, hw, **kw):
    from mpreid.ops import VitEncoder
    kw.setdefault("precision", "fp16")   # these tests pin the fp16-operand mode unless they name another (VitEncoder's default is 'split')
    return VitEncoder(cfg, sd, hw, **kw)


def test_vit_small_vs_reference(golden):
    from mpreid import synth
    g = golden("vit.npz")
    sd = synth.vit_state_dict(SMALL, seed=7, std=0.05, ln_jitter=0.1)
    enc = _encoder(SMALL, sd, (64, 32))
    imgs = synth.synthetic_images(3, 64, 32, seed=3)
    f = enc(torch.from_numpy(imgs)).cpu().numpy()
    assert f.shape == (3, 192)
    _close(f, g["small_feat"])


def test_vit_b16_vs_reference(golden):
    from mpreid import synth
    g = golden("vit.npz")
    big = synth.VIT_B16
    sd = synth.vit_state_dict(big, seed=7, std=0.02, ln_jitter=0.05)
    enc = _encoder(big, sd, (256, 128))
    imgs = synth.synthetic_images(4, 256, 128, seed=1234)
    f = enc(torch.from_numpy(imgs)).cpu().numpy()
    assert f.shape == (4, 1280)
    print("b16 rel-L2 vs reference:", _close(f, g["b16_feat"]))
    f = enc(torch.from_numpy(imgs), cv_emb=torch.from_numpy(g["b16_cv"])).cpu().numpy()
    _close(f, g["b16_feat_cv"])
    # batch independence: same images inside a larger, ragged batch give the same rows bit for bit
    more = synth.synthetic_images(70, 256, 128, seed=99)
    more[10:14] = imgs
    f2 = enc(torch.from_numpy(more)).cpu().numpy()
    f1 = enc(torch.from_numpy(imgs)).cpu().numpy()
    assert np.array_equal(f2[10:14], f1)


def test_vit_stride12_vs_reference(golden):
    from mpreid import synth
    g = golden("vit.npz")
    s12 = dict(synth.VIT_B16, h_res=21, w_res=10, stride=12)
    sd = synth.vit_state_dict(s12, seed=8, std=0.02, ln_jitter=0.05)
    enc = _encoder(s12, sd, (256, 128))
    imgs = synth.synthetic_images(4, 256, 128, seed=1234)[:2]
    _close(enc(torch.from_numpy(imgs)).cpu().numpy(), g["b16_s12_feat"])


def test_vit_bn_neck_vs_oracle():
    from mpreid import synth
    rng = np.random.default_rng(0)
    sd = synth.vit_state_dict(SMALL, seed=11, std=0.05, ln_jitter=0.1)
    bn = dict(bottleneck=(1 + 0.1 * rng.standard_normal(128), 0.1 * rng.standard_normal(128),
                          0.2 * rng.standard_normal(128), 0.5 + rng.random(128)),
              bottleneck_proj=(1 + 0.1 * rng.standard_normal(64), 0.1 * rng.standard_normal(64),
                               0.2 * rng.standard_normal(64), 0.5 + rng.random(64)))
    bn = {k: tuple(np.asarray(a, np.float32) for a in v) for k, v in bn.items()}
    imgs = synth.synthetic_images(5, 64, 32, seed=4)
    enc = _encoder(SMALL, sd, (64, 32), neck_after=True, bn=bn)
    want = orc.vit_features(sd, SMALL, imgs, bn=bn, neck_feat="after")
    _close(enc(torch.from_numpy(imgs)).cpu().numpy(), want)


def test_vit_cls_only_last_block_is_bit_identical():
    """the last block restricted to the CLS row (the only row the output uses) must not change a bit"""
    from mpreid import synth
    big = synth.VIT_B16
    sd = synth.vit_state_dict(big, seed=7, std=0.02, ln_jitter=0.05)
    imgs = torch.from_numpy(synth.synthetic_images(9, 256, 128, seed=5))
    full = _encoder(big, sd, (256, 128), cls_only_last=False)(imgs).cpu().numpy()
    tail = _encoder(big, sd, (256, 128), cls_only_last=True)(imgs).cpu().numpy()
    assert np.array_equal(full, tail)


def test_vit_uint8_input_matches_float_path():
    """uint8 HWC images + fused ToTensor/Normalize == the fp32 entry point fed with the transformed tensor"""
    from mpreid import synth
    rng = np.random.default_rng(0)
    sd = synth.vit_state_dict(SMALL, seed=7, std=0.05, ln_jitter=0.1)
    enc = _encoder(SMALL, sd, (64, 32))
    u8 = rng.integers(0, 256, size=(5, 64, 32, 3), dtype=np.uint8)
    mean, std = (0.5, 0.4, 0.45), (0.5, 0.25, 0.3)
    x = torch.from_numpy(u8).permute(0, 3, 1, 2).float().div(255.0)     # ToTensor
    x = (x - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)  # Normalize
    want = enc(x).cpu().numpy()
    got = enc.forward_u8(torch.from_numpy(u8), mean, std).cpu().numpy()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("name,n,k,epi", [("qkv", 2304, 768, 1), ("out", 768, 768, 2), ("fc1", 3072, 768, 3),
                                            ("fc2", 768, 3072, 2), ("f32", 768, 768, 0),
                                            ("conv1x1_relu", 512, 2048, 7), ("conv1x1_add_relu", 2048, 512, 8)])
def test_gemm_kernels_agree_bitwise(name, n, k, epi):
    """The dispatcher picks the 128x128 or the persistent 256x256 kernel by tile count, so the same image goes
    through either depending on the batch it arrives in: both must give the same bits for every epilogue."""
    import ctypes as C
    from mpreid import _lib
    L = _lib.load()
    dev = _lib.require_gpu()
    gen = torch.Generator(device="cpu").manual_seed(11)
    mb, ms = 16384, 256       # 64 x (n/256) >= 128 big tiles -> persistent kernel; 2 x (n/128) tiles -> small kernel
    a = (torch.rand((mb, k), generator=gen) * 2 - 1).half().to(dev)
    w = ((torch.rand((n, k), generator=gen) * 2 - 1) * 0.05).half().to(dev)
    bias = torch.randn(n, generator=gen).to(dev)
    dt = torch.float16 if epi in (1, 3, 7, 8) else torch.float32
    init = torch.randn((mb, n), generator=gen).to(dt).to(dev)
    ob, osm = init.clone(), init[:ms].clone()
    for x, o in ((a, ob), (a[:ms].contiguous(), osm)):
        _lib.check(L.mpreid_gemm_f16_nt_ex(C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(o.data_ptr()),
                                           C.c_void_p(bias.data_ptr()), x.shape[0], n, k, epi, _lib.stream_ptr()), name)
    torch.cuda.synchronize()
    assert torch.equal(ob[:ms], osm)
    ref = a[:ms].float() @ w.float().T
    if epi in (1, 3):
        ref = ref + bias
        ref = ref * torch.sigmoid(1.702 * ref) if epi == 3 else ref
    elif epi == 2:
        ref = init[:ms].float() + (ref + bias)
    elif epi == 7:
        ref = torch.relu(ref + bias)
    elif epi == 8:
        ref = torch.relu(init[:ms].float() + (ref + bias))
    assert torch.allclose(osm.float(), ref, rtol=2e-3, atol=2e-3)


# ---- `split` precision mode (MODEL.ENCODER_PRECISION: split): fp16 operand PAIRS hi + lo, three products per
# multiply-add on the fp16 matrix cores, fp32 accumulation -- fp32-grade results.  Reference: the encoder runs in fp32
# (processor/processor.py:187-198 has no autocast, model/clip/model.py:654-675 casts the weights to float).
def _pair(x, scale=1.0):
    """fp32 [r, c] -> fp16 pair [r, 2c] = hi | lo of x * scale, through the C ABI"""
    import ctypes as C
    from mpreid import _lib
    x = x.contiguous()
    y = torch.empty((x.shape[0], 2 * x.shape[1]), dtype=torch.float16, device=x.device)
    _lib.check(_lib.load().mpreid_split_pack_f32(C.c_void_p(x.data_ptr()), x.shape[0], x.shape[1], float(scale),
                                                 C.c_void_p(y.data_ptr()), _lib.stream_ptr()), "split_pack")
    return y


def test_split_pack_is_exact_pair():
    from mpreid import _lib
    dev = _lib.require_gpu()
    gen = torch.Generator(device="cpu").manual_seed(3)
    x = (torch.randn((37, 200), generator=gen) * torch.logspace(-6, 2, 200)).to(dev)
    y = _pair(x, 4.0)
    hi, lo = y[:, :200].double(), y[:, 200:].double()
    assert torch.equal(y[:, :200], (x * 4).half())
    # hi + lo reproduces x * scale to 2^-22 relative (or the bottom of the fp16 subnormal range)
    err = (hi + lo - x.double() * 4).abs()
    assert bool((err <= (x.double() * 4).abs() * 2.0 ** -22 + 2.0 ** -25).all())


@pytest.mark.parametrize("name,n,k,epi", [("qkv", 2304, 768, 10), ("out", 768, 768, 11), ("fc1", 3072, 768, 12),
                                            ("fc2", 768, 3072, 11)])
def test_split_gemm_kernels_agree_bitwise_and_are_fp32_grade(name, n, k, epi):
    """128x128 and persistent 256x256 kernel give the same bits in split mode too, and the result is at the fp32
    rounding level of an fp64 reference (the fp16 one-pass GEMM is ~1e-3 away on the same data)"""
    import ctypes as C
    from mpreid import _lib
    L = _lib.load()
    dev = _lib.require_gpu()
    gen = torch.Generator(device="cpu").manual_seed(11)
    mb, ms = 16384, 256
    a = (torch.rand((mb, k), generator=gen) * 2 - 1).to(dev)
    w = ((torch.rand((n, k), generator=gen) * 2 - 1) * 0.05).to(dev)
    e = 9 - int(np.floor(np.log2(float(w.abs().max()))))
    a2, w2 = _pair(a), _pair(w, 2.0 ** e)
    bias = torch.randn(n, generator=gen).to(dev)
    init = torch.randn((mb, n), generator=gen).to(dev)
    if epi == 12:
        ob = torch.empty((mb, 2 * n), dtype=torch.float16, device=dev)
        osm = torch.empty((ms, 2 * n), dtype=torch.float16, device=dev)
    else:
        ob, osm = init.clone(), init[:ms].clone()
    for x, o in ((a2, ob), (a2[:ms].contiguous(), osm)):
        _lib.check(L.mpreid_gemm_f16_split_nt(C.c_void_p(x.data_ptr()), C.c_void_p(w2.data_ptr()), C.c_void_p(o.data_ptr()),
                                              C.c_void_p(bias.data_ptr()), x.shape[0], n, k, float(2.0 ** -e), epi,
                                              _lib.stream_ptr()), name)
    torch.cuda.synchronize()
    assert torch.equal(ob[:ms], osm)
    ref = a[:ms].double() @ w.double().T + bias.double()
    if epi == 11:
        ref = init[:ms].double() + ref
        got = osm.double()
    elif epi == 12:
        ref = ref * torch.sigmoid(1.702 * ref)
        got = osm[:, :n].double() + osm[:, n:].double()
    else:
        got = osm.double()
    rel = float((got - ref).norm() / ref.norm())
    print(f"split gemm {name}: rel-L2 vs fp64 {rel:.2e}, max abs {float((got - ref).abs().max()):.2e}")
    # (fp32 accumulation over 3 * k / 32 matrix instructions: ~1.7e-7 at k = 768, ~4.4e-7 at k = 3072; one-pass fp16: ~3e-4)
    assert rel <= 6e-7 and float((got - ref).abs().max()) <= 4e-6 * max(1.0, float(ref.abs().max()))


def test_split_vit_vs_reference_goldens(golden):
    """the split mode against the reference's own fp32 outputs (tests/golden/vit.npz) at fp32 accuracy: reduced
    config (128x128-tile GEMM kernels, 2-tile attention), ViT-B/16, camera embedding, stride 12 (L = 211)"""
    from mpreid import synth
    g = golden("vit.npz")
    enc = _encoder(SMALL, synth.vit_state_dict(SMALL, seed=7, std=0.05, ln_jitter=0.1), (64, 32), precision="split")
    f = enc(torch.from_numpy(synth.synthetic_images(3, 64, 32, seed=3))).cpu().numpy()
    assert np.abs(f - g["small_feat"]).max() <= 2e-5
    big = synth.VIT_B16
    sd = synth.vit_state_dict(big, seed=7, std=0.02, ln_jitter=0.05)
    imgs = torch.from_numpy(synth.synthetic_images(4, 256, 128, seed=1234))
    enc = _encoder(big, sd, (256, 128), precision="split")
    f = enc(imgs).cpu().numpy()
    print("split b16 max |d| vs reference:", np.abs(f - g["b16_feat"]).max(),
          "rel-L2", np.linalg.norm(f - g["b16_feat"]) / np.linalg.norm(g["b16_feat"]))
    assert np.abs(f - g["b16_feat"]).max() <= 5e-5
    assert np.abs(enc(imgs, cv_emb=torch.from_numpy(g["b16_cv"])).cpu().numpy() - g["b16_feat_cv"]).max() <= 5e-5
    # batch independence (the same image through the 128x128 or the 256x256 kernels, any position): same bits
    more = synth.synthetic_images(70, 256, 128, seed=99)
    more[10:14] = imgs.numpy()
    assert np.array_equal(enc(torch.from_numpy(more)).cpu().numpy()[10:14], f)
    # the last block restricted to the CLS row: same bits
    full = _encoder(big, sd, (256, 128), precision="split", cls_only_last=False)(imgs).cpu().numpy()
    assert np.array_equal(full, f)
    s12 = dict(big, h_res=21, w_res=10, stride=12)
    enc = _encoder(s12, synth.vit_state_dict(s12, seed=8, std=0.02, ln_jitter=0.05), (256, 128), precision="split")
    assert np.abs(enc(imgs[:2]).cpu().numpy() - g["b16_s12_feat"]).max() <= 5e-5


def test_split_vit_uint8_and_views_match_float_path():
    """split mode: uint8 input == transformed fp32 input bit for bit; fused view gather == materialised view tensor"""
    from mpreid import ops, synth
    rng = np.random.default_rng(0)
    sd = synth.vit_state_dict(SMALL, seed=7, std=0.05, ln_jitter=0.1)
    enc = _encoder(SMALL, sd, (64, 32), precision="split")
    u8 = rng.integers(0, 256, size=(5, 64, 32, 3), dtype=np.uint8)
    mean, std = (0.5, 0.4, 0.45), (0.5, 0.25, 0.3)
    x = torch.from_numpy(u8).permute(0, 3, 1, 2).float().div(255.0)
    x = (x - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)
    want = enc(x).cpu().numpy()
    assert np.array_equal(enc.forward_u8(torch.from_numpy(u8), mean, std).cpu().numpy(), want)
    assert np.array_equal(enc.forward_view(x, ops.VIEW_FLIP).cpu().numpy(), enc(torch.flip(x, [3]).contiguous()).cpu().numpy())
    assert np.array_equal(enc.forward_view(x, ops.VIEW_PSEUDO_RGB).cpu().numpy(),
                          enc(x[:, 0:1].repeat(1, 3, 1, 1).contiguous()).cpu().numpy())


def test_fp32_vit_uint8_and_views_match_materialised_tensors():
    """the all-fp32 mode (round 5: mpreid_vit_forward_f32_view): uint8 input and all three test-time-augmentation views inside
    the fp32 patch gather == the tensors the reference materialises (torch ops: ToTensor + Normalize,
    processor/processor_uniprompt_stage2.py:605-633) through the plain fp32 entry point, bit for bit; 70 images = two
    workspace chunks of 64"""
    from mpreid import ops, synth
    rng = np.random.default_rng(1)
    sd = synth.vit_state_dict(SMALL, seed=7, std=0.05, ln_jitter=0.1)
    enc = _encoder(SMALL, sd, (64, 32), precision="fp32")
    u8 = rng.integers(0, 256, size=(70, 64, 32, 3), dtype=np.uint8)
    mean, std = (0.5, 0.4, 0.45), (0.5, 0.25, 0.3)
    x = torch.from_numpy(u8).permute(0, 3, 1, 2).float().div(255.0)
    x = ((x - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)).contiguous()
    views = {ops.VIEW_ORIGINAL: x, ops.VIEW_FLIP: torch.flip(x, [3]).contiguous(),
             ops.VIEW_PSEUDO_IR: x.mean(dim=1, keepdim=True).repeat(1, 3, 1, 1).contiguous(),
             ops.VIEW_PSEUDO_RGB: x[:, 0:1].repeat(1, 3, 1, 1).contiguous()}
    assert np.array_equal(enc.forward_u8(torch.from_numpy(u8), mean, std).cpu().numpy(), enc(x).cpu().numpy())
    for v, t in views.items():
        want = enc(t).cpu().numpy()
        assert np.array_equal(enc.forward_view(x, v).cpu().numpy(), want), v                         # fp32 input, fused view
        assert np.array_equal(enc.forward_view(torch.from_numpy(u8), v, None, mean, std).cpu().numpy(), want), v   # uint8 input


def test_split_vit_bn_neck_vs_oracle():
    from mpreid import synth
    rng = np.random.default_rng(0)
    sd = synth.vit_state_dict(SMALL, seed=11, std=0.05, ln_jitter=0.1)
    bn = dict(bottleneck=(1 + 0.1 * rng.standard_normal(128), 0.1 * rng.standard_normal(128),
                          0.2 * rng.standard_normal(128), 0.5 + rng.random(128)),
              bottleneck_proj=(1 + 0.1 * rng.standard_normal(64), 0.1 * rng.standard_normal(64),
                               0.2 * rng.standard_normal(64), 0.5 + rng.random(64)))
    bn = {k: tuple(np.asarray(a, np.float32) for a in v) for k, v in bn.items()}
    imgs = synth.synthetic_images(5, 64, 32, seed=4)
    enc = _encoder(SMALL, sd, (64, 32), neck_after=True, bn=bn, precision="split")
    want = orc.vit_features(sd, SMALL, imgs, bn=bn, neck_feat="after")
    assert np.abs(enc(torch.from_numpy(imgs)).cpu().numpy() - want).max() <= 3e-5


def test_lnfold_mode_is_gone_loudly():
    """the folded-LayerNorm form of the split mode (round 3, opt-in) was removed in round 4: 0.5 % slower than the plain
    split mode and, at small batches, not reproducible run to run (tools/ws_poison_check.py).  Asking for it fails loudly
    at both levels instead of silently running something else."""
    import ctypes as C
    from mpreid import _lib, ops, synth
    small = dict(h_res=4, w_res=2, patch=16, stride=16, width=128, layers=2, heads=2, out_dim=64)
    sd = synth.vit_state_dict(small, seed=7, std=0.05, ln_jitter=0.1)
    with pytest.raises(ValueError, match="removed"):
        ops.VitEncoder(small, sd, (64, 32), precision="split", ln_fold=True)
    enc = ops.VitEncoder(small, sd, (64, 32), precision="split")
    enc.c_cfg.precision = 2   # what MPREID_VIT_SPLIT_LNFOLD was
    with pytest.raises(RuntimeError, match="removed in round 4"):
        enc(torch.from_numpy(synth.synthetic_images(2, 64, 32, seed=3)))


@pytest.mark.parametrize("n", [16, 24, 37, 508])
def test_encoders_do_not_depend_on_stale_workspace(n):
    """run a batch, poison the cached workspace (NaN pattern, large finite pattern), run again: bit-identical features --
    no kernel reads workspace bytes it did not write, and repeated calls are reproducible, at batch sizes that take the
    128 x 128 kernel for every GEMM (16), a mix (24, 37) and the persistent kernel (508)"""
    from mpreid import ops, synth
    sd = synth.vit_state_dict(synth.VIT_B16, seed=7, std=0.02)
    x = torch.from_numpy(synth.synthetic_images(min(n, 64), 256, 128, seed=n)).cuda()
    x = x.repeat((n + x.shape[0] - 1) // x.shape[0], 1, 1, 1)[:n].contiguous()
    x = x + 0.01 * torch.randn(x.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(n))
    for prec in ("split", "fp16"):
        ops.release_workspaces()
        enc = ops.VitEncoder(synth.VIT_B16, sd, (256, 128), precision=prec, ws_tag="poison")
        ref = enc(x).clone()
        for pat in (0xFF, 0x7B):
            for key, buf in list(ops._ws_cache.items()):
                if key[1] == "poison":
                    buf.fill_(pat)
            assert torch.equal(enc(x), ref), (prec, n, hex(pat))
    ops.release_workspaces("poison")


# ---- the split mode where real CLIP checkpoints will stress it (hard part 5) ---------------------
def outlier_state_dict(cfg, seed=31):
    """seeded ViT weights with the outlier structure of trained CLIP towers (reference model/clip/model.py:150-161 LayerNorm,
    :260-281 block): ONE residual channel carried at about -2000 / +2000 through every block (a 'massive activation'
    channel: written by ln_pre with gamma 100 on a channel that dominates the embedding), LayerNorm gammas with a few
    channels x30-100, and FC1 units whose pre-activations sit at +-60."""
    from mpreid import synth
    sd = synth.vit_state_dict(cfg, seed=seed, std=0.02, ln_jitter=0.05)
    rng = np.random.default_rng(seed)
    w = cfg["width"]
    c0, c1 = 5, w // 2 + 3
    sd["positional_embedding"][:, c0] += 40.0          # dominates every token's embedding ...
    sd["positional_embedding"][:, c1] -= 40.0
    sd["ln_pre.weight"][c0] = 100.0                    # ... ln_pre turns it into ~ +-1900 in the residual stream
    sd["ln_pre.weight"][c1] = 100.0
    for i in range(cfg["layers"]):
        b = f"transformer.resblocks.{i}"
        for ln in ("ln_1", "ln_2"):
            big = rng.choice(w, 4, replace=False)
            big = big[(big != c0) & (big != c1)]
            sd[f"{b}.{ln}.weight"][big] *= rng.uniform(30.0, 100.0, big.size).astype(np.float32)
            sd[f"{b}.{ln}.weight"][[c0, c1]] = 0.02     # trained towers damp the massive channels inside the blocks
        units = rng.choice(4 * w, 6, replace=False)
        sd[f"{b}.mlp.c_fc.bias"][units[:3]] = 60.0
        sd[f"{b}.mlp.c_fc.bias"][units[3:]] = -60.0
    return sd


def test_split_mode_on_clip_like_outliers():
    """fp16 operand PAIRS on activations with massive channels, x30-100 LayerNorm gammas and +-60 FC1 pre-activations:
    features finite and within 2e-5 (relative L2) of the float64 evaluation of the same graph, ViT-B/16 at full size"""
    from mpreid import synth
    from oracle import oracle as orc
    cfg = synth.VIT_B16
    sd = outlier_state_dict(cfg)
    imgs = synth.synthetic_images(3, 256, 128, seed=12)
    want = orc.vit_features(sd, cfg, imgs, dtype="float64")
    assert np.isfinite(want).all()
    # the stress is real: the residual stream carries the massive channel (checked on the fp64 graph's ln_pre output)
    enc = _encoder(cfg, sd, (256, 128), precision="split")
    got = enc(torch.from_numpy(imgs)).cpu().numpy().astype(np.float64)
    assert np.isfinite(got).all()
    rel = np.linalg.norm(got - want) / np.linalg.norm(want)
    f32 = orc.vit_features(sd, cfg, imgs).astype(np.float64)
    rel32 = np.linalg.norm(f32 - want) / np.linalg.norm(want)
    assert rel <= 2e-5, (rel, rel32)
    assert rel <= max(8.0 * rel32, 4e-6), (rel, rel32)   # no worse than a few times what plain fp32 arithmetic loses on this model


@pytest.mark.parametrize("scale,finite", [(1e-3, True), (1.0, True), (30.0, True), (2e5, False)])
def test_split_mode_input_scale_edges(scale, finite):
    """images far outside val_transforms' range: tiny inputs (most `lo` halves of the patch operands are fp16 subnormals or
    zero) stay within the split mode's bound because ln_pre renormalises what the patch GEMM carried with >= 11 bits
    RELATIVE TO THE ROW; inputs whose `hi` halves overflow fp16 (|x| > 65 504) give non-finite features, which
    R1_mAP_eval.compute() refuses loudly (RuntimeError) instead of ranking them"""
    from mpreid import synth
    from oracle import oracle as orc
    from utils.metrics import R1_mAP_eval
    cfg = dict(h_res=4, w_res=2, patch=16, stride=16, width=128, layers=2, heads=2, out_dim=64)
    sd = synth.vit_state_dict(cfg, seed=7, std=0.05, ln_jitter=0.1)
    imgs = synth.synthetic_images(6, 64, 32, seed=3) * np.float32(scale)
    enc = _encoder(cfg, sd, (64, 32), precision="split")
    got = enc(torch.from_numpy(imgs))
    if finite:
        want = orc.vit_features(sd, cfg, imgs, dtype="float64")
        rel = np.linalg.norm(got.cpu().numpy() - want) / np.linalg.norm(want)
        assert np.isfinite(got.cpu().numpy()).all() and rel <= 2e-5, (scale, rel)
        return
    assert not torch.isfinite(got).all()
    ev = R1_mAP_eval(2, feat_norm=True)
    ev.reset()
    ev.update((got, (0, 1, 0, 1, 0, 1), (0,) * 6))
    with pytest.raises(RuntimeError, match="non-finite"):
        ev.compute()


WALK_WORKER = """
import ctypes as C, sys, os
sys.path[:0] = [{root!r}, os.path.join({root!r}, "mp-reid_amd")]
import numpy as np, torch
from mpreid import _lib
L = _lib.load(); dev = _lib.require_gpu()
def pair(x, scale=1.0):
    y = torch.empty((x.shape[0], 2 * x.shape[1]), dtype=torch.float16, device=dev)
    _lib.check(L.mpreid_split_pack_f32(C.c_void_p(x.data_ptr()), x.shape[0], x.shape[1], scale, C.c_void_p(y.data_ptr()), _lib.stream_ptr()), "pack")
    return y
def gemm(a2, w2, out, bias, n, k, epi):
    _lib.check(L.mpreid_gemm_f16_split_nt(C.c_void_p(a2.data_ptr()), C.c_void_p(w2.data_ptr()), C.c_void_p(out.data_ptr()),
                                          C.c_void_p(bias.data_ptr()), a2.shape[0], n, k, 2.0 ** -13, epi, _lib.stream_ptr()), "gemm")
gen = torch.Generator(device="cpu").manual_seed(5)
for (m, n, k, epi) in {shapes!r}:
    a = (torch.rand((m, k), generator=gen) * 2 - 1).to(dev)
    w = ((torch.rand((n, k), generator=gen) * 2 - 1) * 0.05).to(dev)
    bias = torch.randn(n, generator=gen).to(dev)
    a2, w2 = pair(a), pair(w, 2.0 ** 13)
    init = torch.randn((m, n), generator=gen).to(dev)
    big = torch.empty((m, 2 * n), dtype=torch.float16, device=dev) if epi == 12 else init.clone()
    gemm(a2, w2, big, bias, n, k, epi)                       # the persistent 256 x 256 kernel, tile order per MPREID_TUNE
    ref = torch.empty_like(big) if epi == 12 else init.clone()
    for s in range(0, m, 256):                               # the 128 x 128 kernel, 256 rows at a time: no tile walk at all
        blk = ref[s:s + 256].clone()
        gemm(a2[s:s + 256].contiguous(), w2, blk, bias, n, k, epi)
        ref[s:s + 256] = blk
    torch.cuda.synchronize()
    assert torch.equal(big, ref), (m, n, k, epi, os.environ.get("MPREID_TUNE"))
print("WALK OK")
"""


@pytest.mark.parametrize("tune", ["", "gemm_walk=0", "gemm_walk=2", "gemm_walk=3", "gemm_walk=4", "gemm_ragged=0"])
def test_persistent_gemm_tile_orders_are_bit_identical(tmp_path, tune):
    """every tile order of the persistent split GEMM (MPREID_TUNE gemm_walk: auto = column-fastest for the FC2 shape and row
    groups of 4 when 8 do not divide among the XCDs, row-fastest, column-fastest always, groups of 4 / 16 tile rows) writes the
    FULL output the 128 x 128 kernel writes, bit for bit: FC2 / QKV / FC1 shapes at 64 tile rows, out-proj at 224 tile rows
    (groups of 4) and at 96 (neither 8 nor 4 divides among the XCDs: the plain walk), and three ragged row counts on the owned
    walk.  The tuning string is latched per process."""
    import os
    import subprocess
    import sys
    shapes = [(16384, 768, 3072, 11), (16384, 2304, 768, 10), (16384, 3072, 768, 12), (57344, 768, 768, 11), (24576, 768, 768, 11),
              # ragged row counts on the XCD-owned walk (round 6): 254 tile rows (the patch embedding's M), 245 (a 485-image encode
              # group), 185 = 23 groups of 8 + one single row
              (65024, 768, 768, 11), (62720, 2304, 768, 10), (47360, 768, 768, 11)]
    script = tmp_path / "walk_worker.py"
    script.write_text(WALK_WORKER.format(root=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), shapes=shapes))
    env = dict(os.environ)
    env.pop("MPREID_TUNE", None)
    if tune:
        env["MPREID_TUNE"] = tune
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "WALK OK" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
