"""Seeded synthetic inputs shared by the golden-vector script, the tests and bench.py.

Nothing here comes from the reference: the recipes are SURVEY.md §8(d) ("Synthetic inputs").
Everything is generated with numpy's PCG64 ``default_rng`` so that the same seed gives the
same bytes in this container and on the GPU box (same image, same numpy).
"""
from __future__ import annotations

import numpy as np

__all__ = ["clustered_features", "labels_for", "vit_state_dict", "synthetic_images", "identity_images", "VIT_B16", "rn50_state_dict",
           "RN50"]


def clustered_features(n: int, dim: int, sigma: float, seed: int = 1234, per_id: int = 20,
                       normalize: bool = True):
    """Clustered re-id style features: ``n // per_id`` identity centroids ~ N(0, I), each sample is
    its centroid plus ``sigma`` * N(0, I); rows L2-normalised.  Returns (feat fp32 [n, dim], pid int64 [n])."""
    rng = np.random.default_rng(seed)
    n_ids = max(1, n // per_id)
    cent = rng.standard_normal((n_ids, dim)).astype(np.float32)
    pid = rng.integers(0, n_ids, size=n).astype(np.int64)
    x = cent[pid] + np.float32(sigma) * rng.standard_normal((n, dim)).astype(np.float32)
    if normalize:
        x = x / np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-12)
    return np.ascontiguousarray(x, dtype=np.float32), pid


def labels_for(n: int, n_cams: int = 6, seed: int = 99):
    rng = np.random.default_rng(seed)
    return rng.integers(0, n_cams, size=n).astype(np.int64)


# (h_res, w_res, patch, stride, width, layers, heads, out_dim)
VIT_B16 = dict(h_res=16, w_res=8, patch=16, stride=16, width=768, layers=12, heads=12, out_dim=512)


def vit_state_dict(cfg: dict, seed: int = 7, std: float = 0.02, ln_jitter: float = 0.0):
    """Deterministic random-init weights in the CLIP ``VisionTransformer`` state-dict layout
    (key names as in the checkpoint layout listed in SURVEY.md §5 "Checkpoint / resume").

    normal(0, std) for linear/conv weights, small normal biases, LN gamma=1 beta=0 (optionally
    jittered so that gamma/beta handling is actually exercised), class/pos/proj ~ width**-0.5 * N(0,1).
    Returned as a dict of float32 numpy arrays."""
    rng = np.random.default_rng(seed)
    w, L, od = cfg["width"], cfg["layers"], cfg["out_dim"]
    p = cfg["patch"]
    ntok = cfg["h_res"] * cfg["w_res"] + 1
    scale = w ** -0.5

    def nrm(*shape, s=std):
        return (rng.standard_normal(shape) * s).astype(np.float32)

    def ln(prefix, sd):
        sd[prefix + ".weight"] = (1.0 + ln_jitter * rng.standard_normal(w)).astype(np.float32)
        sd[prefix + ".bias"] = (ln_jitter * rng.standard_normal(w)).astype(np.float32)

    sd = {}
    sd["conv1.weight"] = nrm(w, 3, p, p)
    sd["class_embedding"] = nrm(w, s=scale)
    sd["positional_embedding"] = nrm(ntok, w, s=scale)
    ln("ln_pre", sd)
    for i in range(L):
        b = f"transformer.resblocks.{i}"
        sd[b + ".attn.in_proj_weight"] = nrm(3 * w, w)
        sd[b + ".attn.in_proj_bias"] = nrm(3 * w, s=0.01)
        sd[b + ".attn.out_proj.weight"] = nrm(w, w)
        sd[b + ".attn.out_proj.bias"] = nrm(w, s=0.01)
        ln(b + ".ln_1", sd)
        sd[b + ".mlp.c_fc.weight"] = nrm(4 * w, w)
        sd[b + ".mlp.c_fc.bias"] = nrm(4 * w, s=0.01)
        sd[b + ".mlp.c_proj.weight"] = nrm(w, 4 * w)
        sd[b + ".mlp.c_proj.bias"] = nrm(w, s=0.01)
        ln(b + ".ln_2", sd)
    ln("ln_post", sd)
    sd["proj"] = nrm(w, od, s=scale)
    return sd


# CLIP "RN50" visual tower as the reference builds it for a 256x128 input (model/clip/model.py:509-516:
# heads = width * 32 // 64, spacial_dim = h_resolution * w_resolution = 16 * 8)
RN50 = dict(layers=(3, 4, 6, 3), width=64, heads=32, out_dim=1024, h_res=16, w_res=8)


def rn50_state_dict(cfg: dict, seed: int = 11):
    """Deterministic random weights in the ``ModifiedResNet`` state-dict layout (model/clip/model.py:92-148):
    He-scaled conv weights, BatchNorm with non-trivial gamma / beta / running statistics (so that folding them is
    exercised), attention-pool projections.  float32 numpy arrays (+ int64 num_batches_tracked)."""
    rng = np.random.default_rng(seed)
    width, layers = cfg["width"], cfg["layers"]
    sd = {}

    def conv(name, cout, cin, k):
        sd[name + ".weight"] = (rng.standard_normal((cout, cin, k, k)) * np.sqrt(2.0 / (cin * k * k))).astype(np.float32)

    def bn(name, c, gamma=1.0):
        sd[name + ".weight"] = (gamma * (1.0 + 0.1 * rng.standard_normal(c))).astype(np.float32)
        sd[name + ".bias"] = (0.1 * rng.standard_normal(c)).astype(np.float32)
        sd[name + ".running_mean"] = (0.1 * rng.standard_normal(c)).astype(np.float32)
        sd[name + ".running_var"] = (0.5 + rng.random(c)).astype(np.float32)
        sd[name + ".num_batches_tracked"] = np.array(0, dtype=np.int64)

    conv("conv1", width // 2, 3, 3), bn("bn1", width // 2)
    conv("conv2", width // 2, width // 2, 3), bn("bn2", width // 2)
    conv("conv3", width, width // 2, 3), bn("bn3", width)
    inplanes = width
    for li, (planes, blocks, stride) in enumerate(zip((width, width * 2, width * 4, width * 8), layers, (1, 2, 2, 1)), 1):
        for b in range(blocks):
            pre = f"layer{li}.{b}"
            st = stride if b == 0 else 1
            conv(pre + ".conv1", planes, inplanes, 1), bn(pre + ".bn1", planes)
            conv(pre + ".conv2", planes, planes, 3), bn(pre + ".bn2", planes)
            conv(pre + ".conv3", planes * 4, planes, 1), bn(pre + ".bn3", planes * 4, gamma=0.5)
            if st > 1 or inplanes != planes * 4:
                conv(pre + ".downsample.0", planes * 4, inplanes, 1), bn(pre + ".downsample.1", planes * 4)
            inplanes = planes * 4
    e, s_dim = width * 32, cfg["h_res"] * cfg["w_res"]
    sd["attnpool.positional_embedding"] = (rng.standard_normal((s_dim + 1, e)) / e ** 0.5).astype(np.float32)
    for nm, od in (("k_proj", e), ("q_proj", e), ("v_proj", e), ("c_proj", cfg["out_dim"])):
        sd[f"attnpool.{nm}.weight"] = (rng.standard_normal((od, e)) * e ** -0.5).astype(np.float32)
        sd[f"attnpool.{nm}.bias"] = (0.01 * rng.standard_normal(od)).astype(np.float32)
    return sd


def synthetic_images(n: int, h: int, w: int, seed: int = 1234):
    """``randn`` clamped to [-1, 1] — the value range after mean=std=0.5 normalisation. fp32 NCHW."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, 3, h, w)).astype(np.float32)
    return np.clip(x, -1.0, 1.0)


def identity_images(n_ids: int, per_id: int, beta: float, h: int = 256, w: int = 128, seed: int = 5, grid=None):
    """Synthetic re-id images with identity structure: one random template per identity, every image is
    ``clip((1 - beta) * template + beta * noise, -1, 1)``, shuffled.  Returns (fp32 NCHW images, pid int64).
    With the seeded random-init ViT-B/16, beta ~ 0.55-0.6 gives a Euclidean mAP well inside (0.3, 0.9).
    grid=(gh, gw): LOW-FREQUENCY templates -- a gh x gw grid of uniform random colours, each cell h/gh x w/gw pixels -- for a
    convolutional tower, whose pooled features cannot tell two white-noise templates apart (default: white-noise templates)."""
    rng = np.random.default_rng(seed)
    if grid is None:
        tmpl = np.clip(rng.standard_normal((n_ids, 3, h, w)).astype(np.float32), -1.0, 1.0)
    else:
        gh, gw = grid
        assert h % gh == 0 and w % gw == 0, (h, w, grid)
        cells = rng.uniform(-1.0, 1.0, (n_ids, 3, gh, gw)).astype(np.float32)
        tmpl = np.repeat(np.repeat(cells, h // gh, axis=2), w // gw, axis=3)
    pid = np.repeat(np.arange(n_ids, dtype=np.int64), per_id)
    x = np.empty((n_ids * per_id, 3, h, w), np.float32)
    for i, p in enumerate(pid):   # one image at a time: the noise tensor of 2048 images would be 800 MB
        x[i] = np.clip(tmpl[p] * np.float32(1.0 - beta) + np.float32(beta) * rng.standard_normal((3, h, w)).astype(np.float32),
                       -1.0, 1.0)
    perm = rng.permutation(len(pid))
    return x[perm], pid[perm]
