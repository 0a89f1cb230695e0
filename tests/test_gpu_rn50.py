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
