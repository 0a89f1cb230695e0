"""GPU parity of the data-format steps either side of the encoder: the Pillow-exact
Resize of val_transforms and the test-time-augmentation views / aggregation of the Uni-Prompt evaluation."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

SMALL = dict(h_res=4, w_res=2, patch=16, stride=16, width=128, layers=2, heads=2, out_dim=64)


def test_resize_bit_exact_vs_pillow_goldens(golden):
    """integer/byte work: every output byte equals PIL.Image.resize(BILINEAR)'s.

    VERSION CAVEAT (stated, not hidden): tests/golden/resize.npz was produced by the Pillow in this image, 12.2.0; the
    reference pins pillow 10.4.0 (requirements.txt:122), which cannot be installed here (no index access, no wheel in the
    offline wheelhouse), so the two `Resample.c` versions could not be diffed or run side by side.  What the fixture pins
    is the 8-bit two-pass resample as Pillow publishes it: `precompute_coeffs` in IEEE double with the triangle filter
    widened by the down-scale factor, coefficients rounded to 22-bit fixed point (PRECISION_BITS = 32 - 8 - 2), int32
    accumulation from 1 << 21, `clip8` after an arithmetic shift, horizontal pass into an 8-bit intermediate, then the
    vertical pass.  That code path has carried the same constants since the fixed-point resampler was introduced
    (Pillow 3.x) -- to the best of our knowledge 10.4.0 and 12.2.0 agree byte for byte on plain `Image.resize(size,
    BILINEAR)` (no `box`, no `reducing_gap`), but this build has no way to PROVE it: a maintainer with both wheels runs
    tests/golden/make_goldens.py resize under 10.4.0 and compares the npz."""
    from mpreid import ops
    g = golden("resize.npz")
    n = int(g["n"])
    by_size = {}
    for i in range(n):
        by_size.setdefault(g[f"out{i}"].shape[:2], []).append(i)
    for (oh, ow), idx in by_size.items():     # one ragged batch per target size
        got = ops.resize_bilinear_u8([g[f"in{i}"] for i in idx], (oh, ow)).cpu().numpy()
        for j, i in enumerate(idx):
            assert np.array_equal(got[j], g[f"out{i}"]), (i, int((got[j] != g[f"out{i}"]).sum()))


def test_resize_ragged_batch_vs_oracle():
    """a larger ragged batch at the real target size against the C oracle (itself pinned to Pillow)"""
    from mpreid import ops
    rng = np.random.default_rng(3)
    imgs = []
    for _ in range(37):
        h, w = int(rng.integers(20, 400)), int(rng.integers(10, 200))
        imgs.append(rng.integers(0, 256, (h, w, 3), dtype=np.uint8))
    imgs.append(np.zeros((128, 64, 3), np.uint8))
    imgs.append(np.full((128, 64, 3), 255, np.uint8))
    got = ops.resize_bilinear_u8(imgs, (256, 128)).cpu().numpy()
    for j, im in enumerate(imgs):
        assert np.array_equal(got[j], orc.resize_bilinear_u8(im, 256, 128)), j


def test_resize_then_encode_equals_host_resize_then_encode():
    """the resized bytes feed forward_u8 unchanged: same features as resizing on the host first"""
    from mpreid import ops, synth
    sd = synth.vit_state_dict(SMALL, seed=7, std=0.05, ln_jitter=0.1)
    enc = ops.VitEncoder(SMALL, sd, (64, 32))
    rng = np.random.default_rng(9)
    imgs = [rng.integers(0, 256, (int(rng.integers(30, 90)), int(rng.integers(16, 50)), 3), dtype=np.uint8)
            for _ in range(6)]
    dev = ops.resize_bilinear_u8(imgs, (64, 32))
    host = torch.from_numpy(np.stack([orc.resize_bilinear_u8(im, 64, 32) for im in imgs]))
    assert torch.equal(enc.forward_u8(dev).cpu(), enc.forward_u8(host).cpu())


def _close(got, want, rel=4e-3, mx=3e-2):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    rl2 = np.linalg.norm(got - want) / np.linalg.norm(want)
    assert rl2 <= rel and np.abs(got - want).max() <= mx, (rl2, np.abs(got - want).max())


def test_tta_views_vs_reference(golden):
    """each view's features against the reference model run on the reference's own view tensors (fp16-operand
    encoder: stated tolerance rel-L2 <= 4e-3), and the aggregated + normalised query feature"""
    from mpreid import ops, synth
    g = golden("tta.npz")
    sd = synth.vit_state_dict(SMALL, seed=7, std=0.05, ln_jitter=0.1)
    enc = ops.VitEncoder(SMALL, sd, (64, 32))
    img = torch.from_numpy(synth.synthetic_images(5, 64, 32, seed=21))
    for v in range(4):
        _close(enc.forward_view(img, v).cpu().numpy(), g["f32_views"][v])
    _close(enc.forward_tta(img, normalize=False).cpu().numpy(), g["f32_mean"])
    _close(enc.forward_tta(img, normalize=True).cpu().numpy(), g["f32_mean_norm"], mx=3e-3)
    # uint8 input: ToTensor + Normalize (per-channel mean / std) fused in front of the views
    u8 = torch.from_numpy(g["u8_img"])
    mean, std = g["u8_mean"].tolist(), g["u8_std"].tolist()
    for v in range(4):
        _close(enc.forward_view(u8, v, pixel_mean=mean, pixel_std=std).cpu().numpy(), g["u8_views"][v])
    _close(enc.forward_tta(u8, pixel_mean=mean, pixel_std=std).cpu().numpy(), g["u8_mean_norm"], mx=3e-3)


def test_tta_views_equal_materialised_views_bitwise():
    """the fused view gather must give the same bits as encoding a materialised view tensor (what the reference
    does): flip and pseudo-RGB are pure index changes, pseudo-IR is ((c0 + c1) + c2) / 3 in fp32"""
    from mpreid import ops, synth
    sd = synth.vit_state_dict(SMALL, seed=7, std=0.05, ln_jitter=0.1)
    enc = ops.VitEncoder(SMALL, sd, (64, 32))
    img = torch.from_numpy(synth.synthetic_images(7, 64, 32, seed=5))
    mat = {1: torch.flip(img, [3]), 3: img[:, 0:1].repeat(1, 3, 1, 1),
           2: (((img[:, 0] + img[:, 1]) + img[:, 2]) / 3.0).unsqueeze(1).repeat(1, 3, 1, 1)}
    for v, t in mat.items():
        assert torch.equal(enc.forward_view(img, v).cpu(), enc(t.contiguous()).cpu()), v
    assert torch.equal(enc.forward_view(img, 0).cpu(), enc(img).cpu())


def test_pseudo_ir_view_against_a_device_materialised_view():
    """The fused pseudo-IR view follows the HOST arithmetic of `img.mean(dim=1)` -- ((c0 + c1) + c2) / 3 with a correctly
    rounded division, what the reference's CPU path (north_star's parity target) and the goldens of tests/golden/tta.npz
    compute.  torch's DEVICE kernel multiplies the sum by a rounded 1/3 instead: its pixels differ from the host's by at most
    one ulp (on about a third of them), so the features of a device-materialised view are NOT bit-identical to the fused
    view -- they agree to the encoder's own fp32 level.  (Advisor r5: the bit-identity claim holds against the CPU
    materialisation only; this test states the device side with its tolerance.)"""
    from mpreid import ops, synth
    sd = synth.vit_state_dict(SMALL, seed=7, std=0.05, ln_jitter=0.1)
    img = torch.from_numpy(synth.synthetic_images(7, 64, 32, seed=5))
    host_view = img.mean(dim=1, keepdim=True)                     # CPU: sum, then a true division
    dev_view = img.cuda().mean(dim=1, keepdim=True).cpu()         # device kernel: sum * (1 / 3)
    ulp = np.spacing(np.abs(host_view.numpy()).astype(np.float32))
    diff = np.abs(dev_view.numpy() - host_view.numpy())
    assert (diff <= ulp).all(), float((diff / ulp).max())         # at most one ulp apart, pixel by pixel
    for prec, tol in (("split", 2e-6), ("fp32", 2e-6)):
        enc = ops.VitEncoder(SMALL, sd, (64, 32), precision=prec)
        fused = enc.forward_view(img, 2).cpu().numpy()
        assert np.array_equal(fused, enc(host_view.repeat(1, 3, 1, 1).contiguous()).cpu().numpy()), prec   # host arithmetic: same bits
        dev = enc(dev_view.repeat(1, 3, 1, 1).contiguous()).cpu().numpy()
        rel = float(np.linalg.norm(fused - dev) / np.linalg.norm(dev))
        assert rel <= tol, (prec, rel)


def test_tta_mean_bit_exact():
    """stack(...).mean(0) = sequential fp32 sum in view order / n; then the l2_normalize arithmetic of the
    distance path (oracle l2_normalize)"""
    from mpreid import ops
    rng = np.random.default_rng(2)
    f = rng.standard_normal((4, 33, 1280)).astype(np.float32)
    want = ((f[0] + f[1]) + f[2]) + f[3]
    want = (want / np.float32(4)).astype(np.float32)
    got = ops.tta_mean(torch.from_numpy(f), normalize=False).cpu().numpy()
    assert np.array_equal(got, want)
    assert np.array_equal(got, torch.from_numpy(f).mean(0).numpy())       # what the reference computes
    gotn = ops.tta_mean(torch.from_numpy(f), normalize=True).cpu().numpy()
    assert np.array_equal(gotn, orc.l2_normalize(want))
    one = ops.tta_mean(torch.from_numpy(f[:1]), normalize=False).cpu().numpy()   # TTA disabled: a single view
    assert np.array_equal(one, f[0])
