"""Tensor-level host API over the C ABI (torch is plumbing here: device memory + streams).

Everything in this module runs on the current HIP device through libmpreid_hip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from ._lib import GEMM_F16_FAST, GEMM_F16_SPLIT3, GEMM_F32_EXACT  # noqa: F401  (re-exported)
from ._lib import RERANK_AUTO, RERANK_DENSE, RERANK_SPARSE, RERANK_SPARSE_SPLIT3  # noqa: F401

_ws_cache: Dict[tuple, torch.Tensor] = {}


def _workspace(tag: str, nbytes: int, device) -> torch.Tensor:
    """Grow-only cached byte buffer per (device, tag, current stream); the caller owns nothing.  The stream is part of
    the key: a workspace is scratch memory of ONE in-flight call, and calls issued on different streams may overlap on
    the device (processor.do_inference alternates its encoder calls over two streams)."""
    key = (str(device), tag, int(torch.cuda.current_stream(device).cuda_stream))
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            del _ws_cache[key]
            del buf
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def release_workspaces(tag: Optional[str] = None):
    """drop the cached workspace buffers (all of them, or those of one tag on every device)"""
    if tag is None:
        _ws_cache.clear()
        return
    for key in [k for k in _ws_cache if k[1] == tag]:
        del _ws_cache[key]


def _dev_f32(t, device) -> torch.Tensor:
    if isinstance(t, np.ndarray):
        t = torch.from_numpy(t)
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(0 if t is None else t.data_ptr())


def l2_normalize(x: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    """utils/metrics.py:112-114 — F.normalize(feats, dim=1, p=2)."""
    dev = _lib.require_gpu()
    x = _dev_f32(x, dev)
    out = torch.empty_like(x)
    if x.shape[0]:
        _lib.check(_lib.load().mpreid_l2_normalize_f32(_ptr(x), x.shape[0], x.shape[1], eps, _ptr(out),
                                                       _lib.stream_ptr()), "mpreid_l2_normalize_f32")
    return out


def sqnorm(x: torch.Tensor) -> torch.Tensor:
    dev = _lib.require_gpu()
    x = _dev_f32(x, dev)
    out = torch.empty(x.shape[0], dtype=torch.float32, device=dev)
    if x.shape[0]:
        _lib.check(_lib.load().mpreid_sqnorm_f32(_ptr(x), x.shape[0], x.shape[1], _ptr(out), _lib.stream_ptr()),
                   "mpreid_sqnorm_f32")
    return out


def _distance(fn_name: str, q, g, mode: int, out: Optional[torch.Tensor], col_offset: int) -> torch.Tensor:
    dev = _lib.require_gpu()
    L = _lib.load()
    q, g = _dev_f32(q, dev), _dev_f32(g, dev)
    assert q.dim() == 2 and g.dim() == 2 and q.shape[1] == g.shape[1], (q.shape, g.shape)
    nq, ng, d = q.shape[0], g.shape[0], q.shape[1]
    if out is None:
        out = torch.empty((nq, ng), dtype=torch.float32, device=dev)
        col_offset = 0
    assert out.is_contiguous() and out.dtype == torch.float32 and out.shape[0] == nq
    ldo = out.shape[1]
    assert col_offset + ng <= ldo
    if nq == 0 or ng == 0:
        return out
    wsb = L.mpreid_distance_workspace_bytes(nq, ng, d, mode)
    ws = _workspace("distance", wsb, dev)
    optr = C.c_void_p(out.data_ptr() + 4 * col_offset)
    _lib.check(getattr(L, fn_name)(_ptr(q), _ptr(g), nq, ng, d, optr, ldo, mode, _ptr(ws), ws.numel(),
                                   _lib.stream_ptr()), fn_name)
    return out


def euclidean_distance(q, g, mode: int = GEMM_F32_EXACT, out: Optional[torch.Tensor] = None,
                       col_offset: int = 0) -> torch.Tensor:
    """utils/metrics.py:7-13 on the GPU; returns a device tensor [nq, ng] (squared L2)."""
    return _distance("mpreid_euclidean_distance_f32", q, g, mode, out, col_offset)


def cosine_similarity(q, g, mode: int = GEMM_F32_EXACT, out: Optional[torch.Tensor] = None,
                      col_offset: int = 0) -> torch.Tensor:
    """utils/metrics.py:15-25 on the GPU; returns a device tensor [nq, ng] (arccos of the cosine)."""
    return _distance("mpreid_cosine_similarity_f32", q, g, mode, out, col_offset)


def re_ranking(q, g, k1: int, k2: int, lambda_value: float, local_distmat=None, only_local: bool = False,
               timing: bool = False, debug: bool = False, algo: int = _lib.RERANK_AUTO, ws_tag: str = "rerank"):
    """utils/reranking.py:29-100 on the GPU.  Returns (device tensor [nq, ng] fp32, stats dict)
    and, with debug=True, additionally (initial_rank[:, :k1+1], nnz(V) per row, nnz(V_qe) per row).
    algo: RERANK_AUTO (the candidate pipeline without the N x N matrix when it applies, else the dense one; a sparse
    call that hits a data-dependent capacity is repeated densely), RERANK_DENSE, RERANK_SPARSE -- same bits;
    RERANK_SPARSE_SPLIT3: the sparse algorithm with the blend term's distance rows from the fp16 matrix cores (3-term
    split): neighbour table / V / V_qe / Jaccard term bit-identical, |final - exact| <= lambda * 1e-6 / max.
    ws_tag: name of the cached workspace; calls that run CONCURRENTLY on different streams need different tags (the C
    entry point is re-entrant per stream with caller-owned workspaces)."""
    dev = _lib.require_gpu()
    L = _lib.load()
    q, g = _dev_f32(q, dev), _dev_f32(g, dev)
    nq, ng, d = q.shape[0], g.shape[0], q.shape[1]
    N = nq + ng
    loc = None
    if local_distmat is not None:
        loc = _dev_f32(local_distmat, dev)
        assert tuple(loc.shape) == (N, N)
    out = torch.empty((nq, ng), dtype=torch.float32, device=dev)
    st = _lib.RerankStats()
    for attempt in (algo, _lib.RERANK_DENSE):
        wsb = L.mpreid_rerank_workspace_bytes_ex(nq, ng, d, int(k1), int(k2), int(loc is not None), int(attempt))
        ws = _workspace(ws_tag, wsb, dev)
        rc = L.mpreid_rerank_f32_ex(_ptr(q), _ptr(g), nq, ng, d, int(k1), int(k2), float(lambda_value), _ptr(loc),
                                    int(bool(only_local)), _ptr(out), ng, _ptr(ws), ws.numel(), _lib.stream_ptr(),
                                    C.byref(st), int(bool(timing)), int(attempt))
        if rc == _lib.ERR_RETRY_DENSE and attempt != _lib.RERANK_DENSE and algo == _lib.RERANK_AUTO:
            continue
        _lib.check(rc, "mpreid_rerank_f32_ex")
        break
    stats = st.as_dict()
    if not debug:
        return out, stats
    rank = np.empty((N, k1 + 1), np.int32)
    vc = np.empty(N, np.int32)
    vq = np.empty(N, np.int32)
    _lib.check(L.mpreid_rerank_debug_copy_ex(_ptr(ws), nq, ng, d, int(k1), int(k2), int(loc is not None),
                                             C.c_void_p(rank.ctypes.data), C.c_void_p(vc.ctypes.data),
                                             C.c_void_p(vq.ctypes.data), _lib.stream_ptr(), int(stats["algo"])),
               "mpreid_rerank_debug_copy_ex")
    return out, stats, rank, vc, vq


def gemm_f16_nt(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """C[M,N] fp32 = A[M,K] fp16 x B[N,K]^T fp16 (M, N multiples of 128, K of 64)."""
    dev = _lib.require_gpu()
    assert a.dtype == torch.float16 and b.dtype == torch.float16 and a.is_contiguous() and b.is_contiguous()
    m, k = a.shape
    n = b.shape[0]
    c = torch.empty((m, n), dtype=torch.float32, device=dev)
    _lib.check(_lib.load().mpreid_gemm_f16_nt(_ptr(a), _ptr(b), _ptr(c), m, n, k, _lib.stream_ptr()),
               "mpreid_gemm_f16_nt")
    return c


VIEW_ORIGINAL, VIEW_FLIP, VIEW_PSEUDO_IR, VIEW_PSEUDO_RGB = 0, 1, 2, 3


def tta_mean(feats: torch.Tensor, normalize: bool = True) -> torch.Tensor:
    """feats [n_views, rows, dim] fp32 -> [rows, dim]: torch.stack(feat_list).mean(0) (+ F.normalize)."""
    dev = _lib.require_gpu()
    f = _dev_f32(feats, dev)
    assert f.dim() == 3
    out = torch.empty(f.shape[1:], dtype=torch.float32, device=dev)
    _lib.check(_lib.load().mpreid_tta_mean_f32(_ptr(f), f.shape[0], f.shape[1], f.shape[2], int(bool(normalize)), _ptr(out),
                                               _lib.stream_ptr()), "mpreid_tta_mean_f32")
    return out


class _HostStager:
    """Two pinned host buffers used alternately for H2D uploads: packing batch i+1 on the host overlaps the copy of
    batch i, and a buffer is reused only after the copy that read it has finished (event)."""

    def __init__(self, n: int = 2):
        self.bufs, self.events, self.i = [None] * n, [None] * n, 0

    def stage(self, nbytes: int):
        self.i = (self.i + 1) % len(self.bufs)
        i = self.i
        if self.events[i] is not None:
            self.events[i].synchronize()
        if self.bufs[i] is None or self.bufs[i].numel() < nbytes:
            self.bufs[i] = torch.empty(max(int(nbytes * 1.5), 1 << 20), dtype=torch.uint8).pin_memory()
        return self.bufs[i][:nbytes], i

    def copied(self, i: int):
        self.events[i] = torch.cuda.Event()
        self.events[i].record(torch.cuda.current_stream())


_stager = _HostStager()


def raw_image_layout(images):
    """(contiguous uint8 [h, w, 3] arrays, hw int32 [B, 2], byte offsets int64 [B], total bytes) of a sequence of decoded
    RGB images laid back to back"""
    arrs = [np.ascontiguousarray(im.cpu().numpy() if isinstance(im, torch.Tensor) else im, dtype=np.uint8) for im in images]
    assert len(arrs) > 0 and all(a.ndim == 3 and a.shape[2] == 3 for a in arrs)
    hw = np.array([a.shape[:2] for a in arrs], dtype=np.int32)
    sizes = hw[:, 0].astype(np.int64) * hw[:, 1] * 3
    offsets = np.zeros(len(arrs), np.int64)
    offsets[1:] = np.cumsum(sizes)[:-1]
    return arrs, hw, offsets, int(sizes.sum())


class PackedRawImages:
    """A batch of decoded uint8 RGB images of ragged sizes ALREADY on the device, packed back to back: data uint8 [bytes],
    offsets int64 [B], hw int32 [B, 2] (device tensors), max_h = tallest image.  What mpreid.pipeline uploads for a
    RawImageBatch loader; ``resize_packed_u8`` turns it into the uint8 [B, H, W, 3] input of forward_u8."""

    def __init__(self, data, offsets, hw, count, max_h):
        self.data, self.offsets, self.hw, self.count, self.max_h = data, offsets, hw, int(count), int(max_h)

    def __len__(self):
        return self.count


def resize_packed_u8(packed: PackedRawImages, out_hw) -> torch.Tensor:
    """the two resize kernels on images that are already packed in device memory (no host work, no copy)"""
    dev = _lib.require_gpu()
    L = _lib.load()
    B, oh, ow = packed.count, int(out_hw[0]), int(out_hw[1])
    dst = torch.empty((B, oh, ow, 3), dtype=torch.uint8, device=dev)
    ws = _workspace("resize", L.mpreid_resize_workspace_bytes(B, packed.max_h, ow), dev)
    _lib.check(L.mpreid_resize_bilinear_u8(_ptr(packed.data), _ptr(packed.offsets), _ptr(packed.hw), B, packed.max_h, oh, ow,
                                           _ptr(dst), _ptr(ws), ws.numel(), _lib.stream_ptr()), "mpreid_resize_bilinear_u8")
    return dst


def resize_bilinear_u8(images, out_hw) -> torch.Tensor:
    """T.Resize(cfg.INPUT.SIZE_TEST) of val_transforms (datasets/make_dataloader.py:57-58) on the GPU, bit-exact with
    PIL.Image.resize(BILINEAR): images = sequence of uint8 [h, w, 3] arrays / tensors of any sizes (decoded RGB), or a
    PackedRawImages; returns uint8 [B, out_h, out_w, 3] on the device -- the input of VitEncoder.forward_u8.  One H2D
    copy of the packed bytes (pinned staging), two kernels."""
    if isinstance(images, PackedRawImages):
        return resize_packed_u8(images, out_hw)
    dev = _lib.require_gpu()
    arrs, hw, offsets, total = raw_image_layout(images)
    packed, slot = _stager.stage(total)
    np.concatenate([a.reshape(-1) for a in arrs], out=packed.numpy())
    src = packed.to(dev, non_blocking=True)
    _stager.copied(slot)
    return resize_packed_u8(PackedRawImages(src, torch.from_numpy(offsets).to(dev), torch.from_numpy(hw).to(dev),
                                            len(arrs), int(hw[:, 0].max())), out_hw)


# ----------------------------------------------------------------------------------------------
# ViT image encoder
# ----------------------------------------------------------------------------------------------
class VitEncoder:
    """Device-resident CLIP ViT image encoder + feature head.

    cfg keys: h_res, w_res, patch, stride, width, layers, heads, out_dim (mpreid.synth.VIT_B16 layout).
    state_dict: CLIP VisionTransformer key names (optionally prefixed 'image_encoder.'), numpy or torch.
    bn: optional dict(bottleneck=(weight, bias, running_mean, running_var), bottleneck_proj=(...)).
    """

    def __init__(self, cfg: dict, state_dict: dict, img_hw, neck_after: bool = False, bn: Optional[dict] = None,
                 device=None, cls_only_last: bool = True, ws_tag: str = "vit", precision: str = "split",
                 ln_fold: bool = False):
        """precision: 'split' (default: the parity-grade mode, as in config/node.py and bench.py) = every GEMM operand an fp16 pair hi + lo, products hi.hi' + lo.hi' + hi.lo' on the fp16
        matrix cores with fp32 accumulation -- fp32-grade features (~1e-6) at 3x the matrix work: the mode that meets
        the 1e-4 mAP bound AND is the measured one; 'fp16' = fp16 operands, fp32 accumulate / residual stream (fastest,
        relative feature error ~4e-4: misses the bound on hard data); 'fp32' = every weight and activation fp32, exact
        fp32 matrix instruction (~1e-6, ~1/8 of the fp16 throughput; mpreid_vit_forward_f32)."""
        assert precision in ("fp16", "fp32", "split"), precision
        if ln_fold:
            raise ValueError("ln_fold: the folded-LayerNorm form of the split mode was removed in round 4 (0.5 % slower than the plain "
                             "split mode and not reproducible run to run at small batches: include/mpreid.h)")
        self.precision = precision
        self.device = device or _lib.require_gpu()
        self.ws_tag = ws_tag   # encoders that run concurrently on different streams need distinct workspaces
        self.cfg = dict(cfg)
        self.img_hw = tuple(img_hw)
        dev = self.device

        def get(name):
            for k in (name, "image_encoder." + name):
                if k in state_dict:
                    v = state_dict[k]
                    return torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v.detach()
            raise KeyError(name)

        def f32(name):
            return get(name).to(device=dev, dtype=torch.float32).contiguous()

        scales = {}

        def f16(name, shape=None, tensor=None):   # a GEMM weight: fp16 for the MFMA path, fp32 in the all-fp32 mode
            t = (get(name) if tensor is None else tensor).to(device=dev, dtype=torch.float32)
            if shape is not None:
                t = t.reshape(shape)
            if precision == "split":
                # fp16 pair [out][hi(in) | lo(in)] of W * 2^e with the largest |entry| in [2^9, 2^10): hi + lo carries
                # 22 significant bits of every entry that matters; 2^-e is undone in the GEMM epilogue (exact)
                t = t.contiguous()
                amax = float(t.abs().max())
                e = 9 - int(np.floor(np.log2(amax))) if amax > 0 and np.isfinite(amax) else 0
                pair = torch.empty((t.shape[0], 2 * t.shape[1]), dtype=torch.float16, device=dev)
                _lib.check(_lib.load().mpreid_split_pack_f32(_ptr(t), t.shape[0], t.shape[1], float(2.0 ** e), _ptr(pair),
                                                             _lib.stream_ptr()), "mpreid_split_pack_f32")
                scales[name] = float(2.0 ** -e)
                return pair
            return t.contiguous() if precision == "fp32" else t.to(torch.float16).contiguous()

        w = cfg["width"]
        self._keep = []  # owns every device tensor referenced by the C structs
        keep = self._keep.append
        self.c_cfg = _lib.VitCfg(self.img_hw[0], self.img_hw[1], cfg["patch"], cfg["stride"], cfg["h_res"],
                                 cfg["w_res"], w, cfg["layers"], cfg["heads"], cfg["out_dim"], int(bool(neck_after)),
                                 int(bool(cls_only_last)),
                                 _lib.VIT_SPLIT if precision == "split" else _lib.VIT_F16)
        layers = (_lib.VitLayer * max(cfg["layers"], 1))()
        for i in range(cfg["layers"]):
            b = f"transformer.resblocks.{i}"
            t = dict(in_proj_w=f16(b + ".attn.in_proj_weight"), in_proj_b=f32(b + ".attn.in_proj_bias"),
                     out_proj_w=f16(b + ".attn.out_proj.weight"), out_proj_b=f32(b + ".attn.out_proj.bias"),
                     ln1_g=f32(b + ".ln_1.weight"), ln1_b=f32(b + ".ln_1.bias"),
                     ln2_g=f32(b + ".ln_2.weight"), ln2_b=f32(b + ".ln_2.bias"),
                     fc_w=f16(b + ".mlp.c_fc.weight"), fc_b=f32(b + ".mlp.c_fc.bias"),
                     proj_w=f16(b + ".mlp.c_proj.weight"), proj_b=f32(b + ".mlp.c_proj.bias"))
            for k, v in t.items():
                keep(v)
                setattr(layers[i], k, v.data_ptr())
            if precision == "split":
                layers[i].in_proj_s = scales[b + ".attn.in_proj_weight"]
                layers[i].out_proj_s = scales[b + ".attn.out_proj.weight"]
                layers[i].fc_s = scales[b + ".mlp.c_fc.weight"]
                layers[i].proj_s = scales[b + ".mlp.c_proj.weight"]
        self._layers = layers
        top = dict(conv_w=f16("conv1.weight", (w, -1)), class_emb=f32("class_embedding"),
                   pos_emb=f32("positional_embedding"), ln_pre_g=f32("ln_pre.weight"), ln_pre_b=f32("ln_pre.bias"),
                   ln_post_g=f32("ln_post.weight"), ln_post_b=f32("ln_post.bias"), proj=f32("proj"))
        L = cfg["h_res"] * cfg["w_res"] + 1
        assert tuple(top["pos_emb"].shape) == (L, w), (top["pos_emb"].shape, L, w)
        self.c_w = _lib.VitWeights()
        for k, v in top.items():
            keep(v)
            setattr(self.c_w, k, v.data_ptr())
        if precision == "split":
            self.c_w.conv_s = scales["conv1.weight"]
            torch.cuda.current_stream().synchronize()   # the pack kernels read temporaries of this constructor
        if bn is not None:
            for name, (sk, bk) in (("bottleneck", ("bn_scale", "bn_shift")),
                                   ("bottleneck_proj", ("bn_proj_scale", "bn_proj_shift"))):
                wt, bs, mu, var = (torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a).to(
                    device=dev, dtype=torch.float32) for a in bn[name])
                scale = (wt / torch.sqrt(var + 1e-5)).contiguous()
                shift = (bs - mu * scale).contiguous()
                keep(scale), keep(shift)
                setattr(self.c_w, sk, scale.data_ptr())
                setattr(self.c_w, bk, shift.data_ptr())
        self.c_w.layers = C.cast(layers, C.POINTER(_lib.VitLayer))
        self.feat_dim = w + cfg["out_dim"]

    @torch.no_grad()
    def forward(self, img: torch.Tensor, cv_emb: Optional[torch.Tensor] = None,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
        L = _lib.load()
        img = _dev_f32(img, self.device)
        B = img.shape[0]
        assert tuple(img.shape[1:]) == (3,) + self.img_hw, img.shape
        cv = None
        if cv_emb is not None:
            cv = _dev_f32(cv_emb, self.device)
            assert tuple(cv.shape) == (B, self.cfg["width"])
        if out is None:
            out = torch.empty((B, self.feat_dim), dtype=torch.float32, device=self.device)
        assert out.is_contiguous() and out.dtype == torch.float32 and tuple(out.shape) == (B, self.feat_dim)
        if self.precision == "fp32":
            step = 64   # fp32 activations: 4x the bytes per token
            for s in range(0, B, step):
                e = min(B, s + step)
                ws = _workspace(self.ws_tag + "_f32", L.mpreid_vit_workspace_bytes_f32(C.byref(self.c_cfg), e - s), self.device)
                _lib.check(L.mpreid_vit_forward_f32(C.byref(self.c_cfg), C.byref(self.c_w), _ptr(img[s:e]), e - s,
                                                    _ptr(None if cv is None else cv[s:e].contiguous()), _ptr(out[s:e]), _ptr(ws),
                                                    ws.numel(), _lib.stream_ptr()), "mpreid_vit_forward_f32")
            return out
        wsb = L.mpreid_vit_workspace_bytes(C.byref(self.c_cfg), B)
        ws = _workspace(self.ws_tag, wsb, self.device)
        _lib.check(L.mpreid_vit_forward(C.byref(self.c_cfg), C.byref(self.c_w), _ptr(img), B, _ptr(cv), _ptr(out),
                                        _ptr(ws), ws.numel(), _lib.stream_ptr()), "mpreid_vit_forward")
        return out

    __call__ = forward

    def _forward_f32_view(self, img, view, cv_emb, pixel_mean, pixel_std, out):
        """the all-fp32 mode on fp32 [B,3,H,W] or uint8 [B,H,W,3] input, one view: ToTensor + Normalize and the view transform
        inside the patch gather (mpreid_vit_forward_f32_view)"""
        L = _lib.load()
        u8 = img.dtype == torch.uint8
        if u8:
            img = img.detach().to(device=self.device).contiguous()
            assert tuple(img.shape[1:]) == self.img_hw + (3,), img.shape
        else:
            img = _dev_f32(img, self.device)
            assert tuple(img.shape[1:]) == (3,) + self.img_hw, img.shape
        B = img.shape[0]
        cv = None
        if cv_emb is not None:
            cv = _dev_f32(cv_emb, self.device)
            assert tuple(cv.shape) == (B, self.cfg["width"])
        if out is None:
            out = torch.empty((B, self.feat_dim), dtype=torch.float32, device=self.device)
        assert out.is_contiguous() and out.dtype == torch.float32 and tuple(out.shape) == (B, self.feat_dim)
        mean = (C.c_float * 3)(*[float(x) for x in pixel_mean])
        std = (C.c_float * 3)(*[float(x) for x in pixel_std])
        step = 64   # fp32 activations: 4x the bytes per token
        for s in range(0, B, step):
            e = min(B, s + step)
            ws = _workspace(self.ws_tag + "_f32", L.mpreid_vit_workspace_bytes_f32(C.byref(self.c_cfg), e - s), self.device)
            _lib.check(L.mpreid_vit_forward_f32_view(C.byref(self.c_cfg), C.byref(self.c_w), None if u8 else _ptr(img[s:e]),
                                                     _ptr(img[s:e]) if u8 else None, mean, std, int(view), e - s,
                                                     _ptr(None if cv is None else cv[s:e].contiguous()), _ptr(out[s:e]), _ptr(ws),
                                                     ws.numel(), _lib.stream_ptr()), "mpreid_vit_forward_f32_view")
        return out

    @torch.no_grad()
    def forward_u8(self, img_hwc: torch.Tensor, pixel_mean=(0.5, 0.5, 0.5), pixel_std=(0.5, 0.5, 0.5),
                   cv_emb: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """uint8 images [B, H, W, 3] (after Resize); ToTensor + Normalize run inside the patch-gather kernel."""
        if self.precision == "fp32":
            return self._forward_f32_view(img_hwc.to(torch.uint8), VIEW_ORIGINAL, cv_emb, pixel_mean, pixel_std, out)
        L = _lib.load()
        img = img_hwc.detach().to(device=self.device, dtype=torch.uint8).contiguous()
        B = img.shape[0]
        assert tuple(img.shape[1:]) == self.img_hw + (3,), img.shape
        cv = None if cv_emb is None else _dev_f32(cv_emb, self.device)
        if out is None:
            out = torch.empty((B, self.feat_dim), dtype=torch.float32, device=self.device)
        mean = (C.c_float * 3)(*[float(x) for x in pixel_mean])
        std = (C.c_float * 3)(*[float(x) for x in pixel_std])
        wsb = L.mpreid_vit_workspace_bytes(C.byref(self.c_cfg), B)
        ws = _workspace(self.ws_tag, wsb, self.device)
        _lib.check(L.mpreid_vit_forward_u8(C.byref(self.c_cfg), C.byref(self.c_w), _ptr(img), mean, std, B, _ptr(cv),
                                           _ptr(out), _ptr(ws), ws.numel(), _lib.stream_ptr()), "mpreid_vit_forward_u8")
        return out

    @torch.no_grad()
    def forward_view(self, img: torch.Tensor, view: int, cv_emb: Optional[torch.Tensor] = None,
                     pixel_mean=(0.5, 0.5, 0.5), pixel_std=(0.5, 0.5, 0.5),
                     out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One test-time-augmentation view (VIEW_ORIGINAL / VIEW_FLIP / VIEW_PSEUDO_IR / VIEW_PSEUDO_RGB) of a batch:
        img is either fp32 [B,3,H,W] (already normalised) or uint8 [B,H,W,3].  The view transform of
        processor/processor_uniprompt_stage2.py:605-633 happens inside the patch gather."""
        L = _lib.load()
        if self.precision == "fp32":   # (round 5: inside the fp32 patch gather too, no materialised view tensor)
            return self._forward_f32_view(img, view, cv_emb, pixel_mean, pixel_std, out)
        u8 = img.dtype == torch.uint8
        if u8:
            img = img.detach().to(device=self.device).contiguous()
            assert tuple(img.shape[1:]) == self.img_hw + (3,), img.shape
        else:
            img = _dev_f32(img, self.device)
            assert tuple(img.shape[1:]) == (3,) + self.img_hw, img.shape
        B = img.shape[0]
        cv = None if cv_emb is None else _dev_f32(cv_emb, self.device)
        if out is None:
            out = torch.empty((B, self.feat_dim), dtype=torch.float32, device=self.device)
        assert out.is_contiguous() and out.dtype == torch.float32 and tuple(out.shape) == (B, self.feat_dim)
        mean = (C.c_float * 3)(*[float(x) for x in pixel_mean])
        std = (C.c_float * 3)(*[float(x) for x in pixel_std])
        ws = _workspace(self.ws_tag, L.mpreid_vit_workspace_bytes(C.byref(self.c_cfg), B), self.device)
        _lib.check(L.mpreid_vit_forward_view(C.byref(self.c_cfg), C.byref(self.c_w), None if u8 else _ptr(img),
                                             _ptr(img) if u8 else None, mean, std, int(view), B, _ptr(cv), _ptr(out),
                                             _ptr(ws), ws.numel(), _lib.stream_ptr()), "mpreid_vit_forward_view")
        return out

    @torch.no_grad()
    def forward_tta(self, img: torch.Tensor, cv_emb: Optional[torch.Tensor] = None, views=(0, 1, 2, 3),
                    normalize: bool = True, pixel_mean=(0.5, 0.5, 0.5), pixel_std=(0.5, 0.5, 0.5)) -> torch.Tensor:
        """processor/processor_uniprompt_stage2.py:598-640: features of the views, averaged, then L2-normalised when
        TEST.FEAT_NORM is set."""
        B = img.shape[0]
        feats = torch.empty((len(views), B, self.feat_dim), dtype=torch.float32, device=self.device)
        for i, v in enumerate(views):
            self.forward_view(img, v, cv_emb, pixel_mean, pixel_std, out=feats[i])
        return tta_mean(feats, normalize)

    def clone_for_stream(self, ws_tag: str) -> "VitEncoder":
        """a second handle on the same device weights with its own workspace (for a second HIP stream)"""
        import copy
        other = copy.copy(self)
        other.ws_tag = ws_tag
        return other


# ----------------------------------------------------------------------------------------------
# RN50 image encoder (CLIP ModifiedResNet, model/clip/model.py:10-148)
# ----------------------------------------------------------------------------------------------
def _pad_to(n: int, m: int) -> int:
    return (n + m - 1) // m * m


def _np64(a):
    return (a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)).astype(np.float64)


def fold_conv_bn(weight, bn=None, cin_pad: Optional[int] = None, eps: float = 1e-5):
    """Conv2d weight [cout, cin, k, k] (+ eval-mode BatchNorm2d (gamma, beta, running_mean, running_var)) ->
    (fp16 [cout_pad128, k*k*cin_pad] with k order (kh, kw, c), fp32 bias [cout_pad128]); model/clip/model.py:17-24:
    y = gamma * (conv(x) - mean) / sqrt(var + eps) + beta = conv'(x) + b'.  Folded in fp64, rounded once."""
    w = np.asarray(weight, dtype=np.float64)
    cout, cin, kh, kw = w.shape
    if bn is not None:
        gamma, beta, mean, var = (np.asarray(a, dtype=np.float64) for a in bn)
        scale = gamma / np.sqrt(var + eps)
        w = w * scale[:, None, None, None]
        b = beta - mean * scale
    else:
        b = np.zeros(cout)
    cpad = cin if cin_pad is None else cin_pad
    npad = _pad_to(cout, 128)
    wk = np.zeros((npad, kh, kw, cpad), dtype=np.float32)
    wk[:cout, :, :, :cin] = w.transpose(0, 2, 3, 1)
    bk = np.zeros(npad, dtype=np.float32)
    bk[:cout] = b
    return (torch.from_numpy(wk.reshape(npad, kh * kw * cpad)).to(torch.float16), torch.from_numpy(bk))


_zero_pages = {}


def _zero_page(dev):
    if dev not in _zero_pages:
        _zero_pages[dev] = torch.zeros(256, dtype=torch.uint8, device=dev)
    return _zero_pages[dev]


def conv_f16_nhwc(act: torch.Tensor, wgt: torch.Tensor, bias: torch.Tensor, cout: int, taps: int,
                  identity: Optional[torch.Tensor] = None, relu: bool = True) -> torch.Tensor:
    """one folded conv layer on NHWC fp16 [B,H,W,Cin] -> [B,H,W,cout] (stride 1; taps 1 or 9 with pad 1)"""
    dev = _lib.require_gpu()
    assert act.dtype == torch.float16 and act.is_contiguous() and act.is_cuda and wgt.dtype == torch.float16
    B, H, W, C = act.shape
    assert wgt.shape[1] == taps * C and wgt.shape[0] % 128 == 0 and bias.numel() == wgt.shape[0]
    out = torch.empty((B, H, W, cout), dtype=torch.float16, device=dev)
    if identity is not None:
        assert identity.dtype == torch.float16 and identity.is_contiguous() and tuple(identity.shape) == tuple(out.shape)
    _lib.check(_lib.load().mpreid_conv_f16_nhwc(_ptr(act), B, H, W, C, _ptr(wgt), _ptr(bias), cout, wgt.shape[0], taps,
                                                _ptr(identity), int(relu), _ptr(out), _ptr(_zero_page(dev)),
                                                _lib.stream_ptr()), "mpreid_conv_f16_nhwc")
    return out


class Rn50Encoder:
    """Device-resident CLIP RN50 image encoder + the RN50 eval head of build_transformer.

    cfg keys: layers (4-tuple), width, heads, out_dim, h_res, w_res (mpreid.synth.RN50 layout);
    state_dict: ModifiedResNet key names (optionally prefixed 'image_encoder.'), numpy or torch;
    bn: optional dict(bottleneck=(w, b, mean, var), bottleneck_proj=(...)) applied when neck_after.
    Every BatchNorm2d is folded into its convolution here, once (fold_conv_bn); channel counts that are not
    multiples of 64 are stored zero-padded to 64 (the 32-channel stem; reduced test configurations)."""

    def __init__(self, cfg: dict, state_dict: dict, img_hw, neck_after: bool = False, bn: Optional[dict] = None,
                 device=None, ws_tag: str = "rn50", precision: str = "split"):
        """precision: 'fp16' = fp16 NHWC activations, implicit-GEMM convolutions on the fp16 matrix cores (the throughput
        path, relative feature error 2.6e-3); 'fp32' = everything fp32 on the exact fp32 matrix instruction
        (mpreid_rn50_forward_f32: ~1e-6); 'split' = fp32 activations, the convolutions of layer1-4 and the attention pool's
        k / v projections over fp16 PAIRS on the fp16 matrix cores (mpreid_rn50_forward_split: fp32-grade features -- the
        parity-grade mode that is also fast)."""
        assert precision in ("fp16", "fp32", "split"), precision
        self.precision = precision
        self.device = dev = device or _lib.require_gpu()
        self.ws_tag, self.cfg, self.img_hw = ws_tag, dict(cfg), tuple(img_hw)
        width, layers = cfg["width"], tuple(cfg["layers"])
        assert self.img_hw[0] // 16 == cfg["h_res"] and self.img_hw[1] // 16 == cfg["w_res"], (img_hw, cfg)
        if precision in ("fp32", "split"):
            self._init_f32(cfg, state_dict, neck_after, bn, split=precision == "split")
            return

        def get(name):
            for k in (name, "image_encoder." + name):
                if k in state_dict:
                    v = state_dict[k]
                    return v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            raise KeyError(name)

        def bn_of(prefix):
            return tuple(get(f"{prefix}.{a}") for a in ("weight", "bias", "running_mean", "running_var"))

        self._keep = []

        def conv(cname, bname, taps):
            w = get(cname + ".weight")
            cout, cin = w.shape[0], w.shape[1]
            cs_in, cs_out = _pad_to(cin, 64), _pad_to(cout, 64)
            wk, bk = fold_conv_bn(w, bn_of(bname), cin_pad=cs_in)
            if wk.shape[0] < _pad_to(cs_out, 128):  # never: fold pads cout to 128 already
                raise AssertionError
            wk, bk = wk.to(dev).contiguous(), bk.to(dev).contiguous()
            self._keep += [wk, bk]
            return _lib.Rn50Conv(_ptr(wk), _ptr(bk), cs_in, cs_out, wk.shape[0], taps)

        # stem conv1 + bn1 folded, fp32, original [cout][c][kh][kw] order
        g1, b1, m1, v1 = (a.astype(np.float64) for a in bn_of("bn1"))
        sc = g1 / np.sqrt(v1 + 1e-5)
        s1w = torch.from_numpy((get("conv1.weight").astype(np.float64) * sc[:, None, None, None]).astype(np.float32)).to(dev)
        s1b = torch.from_numpy((b1 - m1 * sc).astype(np.float32)).to(dev)
        self._keep += [s1w, s1b]
        blocks = []
        inplanes = width
        for li, (planes, nb, stride) in enumerate(zip((width, width * 2, width * 4, width * 8), layers, (1, 2, 2, 1)), 1):
            for b in range(nb):
                pre = f"layer{li}.{b}"
                blk = _lib.Rn50Block()
                blk.conv1 = conv(pre + ".conv1", pre + ".bn1", 1)
                blk.conv2 = conv(pre + ".conv2", pre + ".bn2", 9)
                blk.conv3 = conv(pre + ".conv3", pre + ".bn3", 1)
                blk.stride = stride if b == 0 else 1
                if blk.stride > 1 or inplanes != planes * 4:
                    blk.down = conv(pre + ".downsample.0", pre + ".downsample.1", 1)
                blocks.append(blk)
                inplanes = planes * 4
        self.c_blocks = (_lib.Rn50Block * len(blocks))(*blocks)
        E, od = width * 32, cfg["out_dim"]
        self.feat_dim = E + od

        def lin(names, pad_rows=None):
            w = np.concatenate([get(f"attnpool.{n}.weight") for n in names]).astype(np.float32)
            b = np.concatenate([get(f"attnpool.{n}.bias") for n in names]).astype(np.float32)
            if pad_rows and w.shape[0] < pad_rows:
                w = np.concatenate([w, np.zeros((pad_rows - w.shape[0], w.shape[1]), np.float32)])
                b = np.concatenate([b, np.zeros(pad_rows - b.shape[0], np.float32)])
            wt, bt = torch.from_numpy(w).to(torch.float16).to(dev), torch.from_numpy(b).to(dev)
            self._keep += [wt, bt]
            return wt, bt

        vw, vb = lin(("v_proj",))
        # k_proj is used transposed (u_h = Wk_h^T q_h, see csrc/rn50.hip); its bias shifts every score of a head by the
        # same amount and cancels in the softmax
        ktw = torch.from_numpy(np.ascontiguousarray(get("attnpool.k_proj.weight").astype(np.float32).T)).to(
            torch.float16).to(dev)
        self._keep.append(ktw)
        qw, qb = lin(("q_proj",))
        cw, cb = lin(("c_proj",), pad_rows=_pad_to(od, 128))
        pos = torch.from_numpy(get("attnpool.positional_embedding").astype(np.float32)).to(dev)
        self._keep.append(pos)
        scale = shift = None
        if neck_after:
            assert bn is not None
            parts = []
            for name in ("bottleneck", "bottleneck_proj"):
                w_, b_, m_, v_ = (_np64(a) for a in bn[name])
                s_ = w_ / np.sqrt(v_ + 1e-5)
                parts.append((s_, b_ - m_ * s_))
            scale = torch.from_numpy(np.concatenate([p[0] for p in parts]).astype(np.float32)).to(dev)
            shift = torch.from_numpy(np.concatenate([p[1] for p in parts]).astype(np.float32)).to(dev)
            self._keep += [scale, shift]
        self.c_cfg = _lib.Rn50Cfg(self.img_hw[0], self.img_hw[1], width, len(blocks), cfg["heads"], od)
        self.c_w = _lib.Rn50Weights()
        self.c_w.stem1_w, self.c_w.stem1_b = _ptr(s1w), _ptr(s1b)
        self.c_w.stem2, self.c_w.stem3 = conv("conv2", "bn2", 9), conv("conv3", "bn3", 9)
        self.c_w.blocks = C.cast(self.c_blocks, C.POINTER(_lib.Rn50Block))
        self.c_w.pos_emb = _ptr(pos)
        self.c_w.kt_w, self.c_w.v_w, self.c_w.v_b = _ptr(ktw), _ptr(vw), _ptr(vb)
        self.c_w.q_w, self.c_w.q_b = _ptr(qw), _ptr(qb)
        self.c_w.c_w, self.c_w.c_b = _ptr(cw), _ptr(cb)
        self.c_w.bn_scale, self.c_w.bn_shift = _ptr(scale), _ptr(shift)

    def _init_f32(self, cfg, state_dict, neck_after, bn, split=False):
        """fp32 mode: BatchNorm folded in fp64 and rounded once to fp32; real channel counts; [cout][kh][kw][cin] rows.
        split: the same folded fp32 matrices, those of layer1-4 and of k_proj / v_proj additionally as fp16 pair matrices
        [cout_pad128][hi(kseg) | lo(kseg)] of W * 2^e (mpreid_split_pack_f32; include/mpreid.h mpreid_rn50_conv_split)"""
        dev = self.device
        width, layers = cfg["width"], tuple(cfg["layers"])

        def get(name):
            for k in (name, "image_encoder." + name):
                if k in state_dict:
                    v = state_dict[k]
                    return v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            raise KeyError(name)

        self._keep = []

        def dev32(a):
            t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
            self._keep.append(t)
            return t

        def fold(cname, bname):
            w = get(cname + ".weight").astype(np.float64)
            g, b, m, v = (get(f"{bname}.{a}").astype(np.float64) for a in ("weight", "bias", "running_mean", "running_var"))
            sc = g / np.sqrt(v + 1e-5)
            return w * sc[:, None, None, None], b - m * sc

        def conv(cname, bname):
            w, b = fold(cname, bname)
            cout, cin, kh, kw = w.shape
            wk = dev32(w.transpose(0, 2, 3, 1).reshape(cout, kh * kw * cin))   # k order (kh, kw, c)
            return _lib.Rn50ConvF32(_ptr(wk), _ptr(dev32(b)), cin, cout, kh * kw)

        def pairs_of(w2d, bias, cin, taps):
            """folded fp32 matrix [cout][taps * cin] + bias -> Rn50ConvSplit (zero-padded pair matrix of W * 2^e)"""
            cout, k = w2d.shape
            kseg, npad = _pad_to(k, 64), _pad_to(cout, 128)
            wp = np.zeros((npad, kseg), np.float32)
            wp[:cout, :k] = w2d.astype(np.float32)
            bp = np.zeros(npad, np.float32)
            bp[:cout] = bias
            amax = float(np.abs(wp).max())
            e = 9 - int(np.floor(np.log2(amax))) if amax > 0 and np.isfinite(amax) else 0
            wt = torch.from_numpy(wp).to(dev)
            pair = torch.empty((npad, 2 * kseg), dtype=torch.float16, device=dev)
            _lib.check(_lib.load().mpreid_split_pack_f32(_ptr(wt), npad, kseg, float(2.0 ** e), _ptr(pair), _lib.stream_ptr()),
                       "mpreid_split_pack_f32")
            torch.cuda.current_stream().synchronize()   # wt is a temporary of this call
            self._keep.append(pair)
            return _lib.Rn50ConvSplit(_ptr(pair), _ptr(dev32(bp)), cin, cout, taps, kseg, npad, float(2.0 ** -e))

        def pairs_of_3x3(w, bias):
            """folded [cout][cin][3][3] -> Rn50ConvSplit for the implicit GEMM (csrc/conv_f16.hip, pair form): fp16 slabs
            [cout_pad128][tap][cin_pad64 / 64][hi(64) | lo(64)] of W * 2^e; kseg = cin_pad64"""
            cout, cin = w.shape[:2]
            cp, npad = _pad_to(cin, 64), _pad_to(cout, 128)
            wp = np.zeros((npad, 9, cp), np.float32)
            wp[:cout, :, :cin] = w.transpose(0, 2, 3, 1).reshape(cout, 9, cin).astype(np.float32)
            bp = np.zeros(npad, np.float32)
            bp[:cout] = bias
            amax = float(np.abs(wp).max())
            e = 9 - int(np.floor(np.log2(amax))) if amax > 0 and np.isfinite(amax) else 0
            wt = torch.from_numpy(wp).to(dev) * float(2.0 ** e)          # exact: a power of two
            hi = wt.half()
            lo = (wt - hi.float()).half()
            hi, lo = hi.view(npad, 9, cp // 64, 1, 64), lo.view(npad, 9, cp // 64, 1, 64)
            slab = torch.cat([hi, lo], dim=3).reshape(npad, 9 * (cp // 64) * 2 * 64).contiguous()
            self._keep.append(slab)
            return _lib.Rn50ConvSplit(_ptr(slab), _ptr(dev32(bp)), cin, cout, 9, cp, npad, float(2.0 ** -e))

        def conv_s(cname, bname):
            w, b = fold(cname, bname)
            cout, cin, kh, kw = w.shape
            if kh * kw == 9:
                return pairs_of_3x3(w, b)
            return pairs_of(w.transpose(0, 2, 3, 1).reshape(cout, kh * kw * cin), b, cin, kh * kw)

        s1w, s1b = fold("conv1", "bn1")
        blocks = []
        inplanes = width
        for li, (planes, nb, stride) in enumerate(zip((width, width * 2, width * 4, width * 8), layers, (1, 2, 2, 1)), 1):
            for b in range(nb):
                pre = f"layer{li}.{b}"
                blk = _lib.Rn50BlockF32()
                blk.conv1, blk.conv2, blk.conv3 = conv(pre + ".conv1", pre + ".bn1"), conv(pre + ".conv2", pre + ".bn2"), \
                    conv(pre + ".conv3", pre + ".bn3")
                blk.stride = stride if b == 0 else 1
                if blk.stride > 1 or inplanes != planes * 4:
                    blk.down = conv(pre + ".downsample.0", pre + ".downsample.1")
                blocks.append(blk)
                inplanes = planes * 4
        self.c_blocks = (_lib.Rn50BlockF32 * len(blocks))(*blocks)
        E, od = width * 32, cfg["out_dim"]
        self.feat_dim = E + od
        self.c_cfg = _lib.Rn50Cfg(self.img_hw[0], self.img_hw[1], width, len(blocks), cfg["heads"], od)
        cw = self.c_w = _lib.Rn50WeightsF32()
        if split:
            sblocks = []
            inpl = width
            for li, (planes, nb, stride) in enumerate(zip((width, width * 2, width * 4, width * 8), layers, (1, 2, 2, 1)), 1):
                for b in range(nb):
                    pre = f"layer{li}.{b}"
                    sb = _lib.Rn50BlockSplit()
                    sb.conv1, sb.conv2, sb.conv3 = conv_s(pre + ".conv1", pre + ".bn1"), conv_s(pre + ".conv2", pre + ".bn2"), \
                        conv_s(pre + ".conv3", pre + ".bn3")
                    sb.stride = stride if b == 0 else 1
                    if sb.stride > 1 or inpl != planes * 4:
                        sb.down = conv_s(pre + ".downsample.0", pre + ".downsample.1")
                    sblocks.append(sb)
                    inpl = planes * 4
            self.c_sblocks = (_lib.Rn50BlockSplit * len(sblocks))(*sblocks)
            self.c_ws = _lib.Rn50WeightsSplit()
            cw = self.c_ws.f32
            self.c_ws.blocks = C.cast(self.c_sblocks, C.POINTER(_lib.Rn50BlockSplit))
            self.c_ws.stem2, self.c_ws.stem3 = conv_s("conv2", "bn2"), conv_s("conv3", "bn3")
            for n in ("k", "v"):
                setattr(self.c_ws, n, pairs_of(get(f"attnpool.{n}_proj.weight").astype(np.float64),
                                               get(f"attnpool.{n}_proj.bias").astype(np.float64), E, 1))
        cw.stem1_w, cw.stem1_b = _ptr(dev32(s1w)), _ptr(dev32(s1b))
        cw.stem2, cw.stem3 = conv("conv2", "bn2"), conv("conv3", "bn3")
        cw.blocks = C.cast(self.c_blocks, C.POINTER(_lib.Rn50BlockF32))
        cw.pos_emb = _ptr(dev32(get("attnpool.positional_embedding")))
        for n in ("q", "k", "v", "c"):
            setattr(cw, n + "_w", _ptr(dev32(get(f"attnpool.{n}_proj.weight"))))
            setattr(cw, n + "_b", _ptr(dev32(get(f"attnpool.{n}_proj.bias"))))
        if neck_after:
            assert bn is not None
            parts = []
            for name in ("bottleneck", "bottleneck_proj"):
                w_, b_, m_, v_ = (_np64(a) for a in bn[name])
                s_ = w_ / np.sqrt(v_ + 1e-5)
                parts.append((s_, b_ - m_ * s_))
            cw.bn_scale = _ptr(dev32(np.concatenate([p[0] for p in parts])))
            cw.bn_shift = _ptr(dev32(np.concatenate([p[1] for p in parts])))

    @torch.no_grad()
    def forward(self, img: torch.Tensor, cv_emb=None, pixel_mean=(0.5, 0.5, 0.5), pixel_std=(0.5, 0.5, 0.5),
                out: Optional[torch.Tensor] = None, view: int = 0) -> torch.Tensor:
        """img: fp32 [B,3,H,W] (val_transforms applied) or uint8 [B,H,W,3] (after Resize).  cv_emb is ignored: the
        reference's RN50 branch has no SIE embedding (model/make_model.py:82-86).  view (split / fp32 towers): a
        test-time-augmentation view applied inside the stem's first convolution (mpreid_rn50_forward_*_view)."""
        L = _lib.load()
        assert view == 0 or self.precision in ("fp32", "split"), "the fp16 tower takes materialised view tensors"
        if self.precision in ("fp32", "split"):
            split = self.precision == "split"
            u8 = img.dtype == torch.uint8    # ToTensor + Normalize inside the stem's first convolution (mpreid_rn50_forward_*_u8)
            if u8:
                img = img.detach().to(device=self.device).contiguous()
                assert tuple(img.shape[1:]) == self.img_hw + (3,), img.shape
                mean = (C.c_float * 3)(*[float(x) for x in pixel_mean])
                std = (C.c_float * 3)(*[float(x) for x in pixel_std])
            else:
                img = _dev_f32(img, self.device)
                assert tuple(img.shape[1:]) == (3,) + self.img_hw, img.shape
            B = img.shape[0]
            if out is None:
                out = torch.empty((B, self.feat_dim), dtype=torch.float32, device=self.device)
            step = 256 if split else 64    # fp32 activations and the im2col matrix: 4x-36x the bytes per image
            for s0 in range(0, B, step):
                e0 = min(B, s0 + step)
                if split:
                    ws = _workspace(self.ws_tag + "_split", L.mpreid_rn50_workspace_bytes_split(C.byref(self.c_cfg), e0 - s0), self.device)
                    if view:
                        _lib.check(L.mpreid_rn50_forward_split_view(C.byref(self.c_cfg), C.byref(self.c_ws), None if u8 else _ptr(img[s0:e0]),
                                                                    _ptr(img[s0:e0]) if u8 else None, mean if u8 else None,
                                                                    std if u8 else None, int(view), e0 - s0, _ptr(out[s0:e0]), _ptr(ws),
                                                                    ws.numel(), _lib.stream_ptr()), "mpreid_rn50_forward_split_view")
                    elif u8:
                        _lib.check(L.mpreid_rn50_forward_split_u8(C.byref(self.c_cfg), C.byref(self.c_ws), _ptr(img[s0:e0]), mean, std,
                                                                  e0 - s0, _ptr(out[s0:e0]), _ptr(ws), ws.numel(), _lib.stream_ptr()),
                                   "mpreid_rn50_forward_split_u8")
                    else:
                        _lib.check(L.mpreid_rn50_forward_split(C.byref(self.c_cfg), C.byref(self.c_ws), _ptr(img[s0:e0]), e0 - s0,
                                                               _ptr(out[s0:e0]), _ptr(ws), ws.numel(), _lib.stream_ptr()),
                                   "mpreid_rn50_forward_split")
                    continue
                ws = _workspace(self.ws_tag + "_f32", L.mpreid_rn50_workspace_bytes_f32(C.byref(self.c_cfg), e0 - s0), self.device)
                if view:
                    _lib.check(L.mpreid_rn50_forward_f32_view(C.byref(self.c_cfg), C.byref(self.c_w), None if u8 else _ptr(img[s0:e0]),
                                                              _ptr(img[s0:e0]) if u8 else None, mean if u8 else None, std if u8 else None,
                                                              int(view), e0 - s0, _ptr(out[s0:e0]), _ptr(ws), ws.numel(),
                                                              _lib.stream_ptr()), "mpreid_rn50_forward_f32_view")
                elif u8:
                    _lib.check(L.mpreid_rn50_forward_f32_u8(C.byref(self.c_cfg), C.byref(self.c_w), _ptr(img[s0:e0]), mean, std,
                                                            e0 - s0, _ptr(out[s0:e0]), _ptr(ws), ws.numel(), _lib.stream_ptr()),
                               "mpreid_rn50_forward_f32_u8")
                else:
                    _lib.check(L.mpreid_rn50_forward_f32(C.byref(self.c_cfg), C.byref(self.c_w), _ptr(img[s0:e0]), e0 - s0,
                                                         _ptr(out[s0:e0]), _ptr(ws), ws.numel(), _lib.stream_ptr()),
                               "mpreid_rn50_forward_f32")
            return out
        u8 = img.dtype == torch.uint8
        if u8:
            img = img.detach().to(device=self.device).contiguous()
            assert tuple(img.shape[1:]) == self.img_hw + (3,), img.shape
        else:
            img = _dev_f32(img, self.device)
            assert tuple(img.shape[1:]) == (3,) + self.img_hw, img.shape
        B = img.shape[0]
        if out is None:
            out = torch.empty((B, self.feat_dim), dtype=torch.float32, device=self.device)
        mean = (C.c_float * 3)(*[float(x) for x in pixel_mean])
        std = (C.c_float * 3)(*[float(x) for x in pixel_std])
        ws = _workspace(self.ws_tag, L.mpreid_rn50_workspace_bytes(C.byref(self.c_cfg), B), self.device)
        _lib.check(L.mpreid_rn50_forward(C.byref(self.c_cfg), C.byref(self.c_w), None if u8 else _ptr(img),
                                         _ptr(img) if u8 else None, mean, std, B, _ptr(out), _ptr(ws), ws.numel(),
                                         _lib.stream_ptr()), "mpreid_rn50_forward")
        return out

    __call__ = forward

    def forward_u8(self, img_hwc, pixel_mean=(0.5, 0.5, 0.5), pixel_std=(0.5, 0.5, 0.5), cv_emb=None, out=None):
        return self.forward(img_hwc, None, pixel_mean, pixel_std, out)

    def forward_view(self, img, view, cv_emb=None, pixel_mean=(0.5, 0.5, 0.5), pixel_std=(0.5, 0.5, 0.5), out=None):
        """one test-time-augmentation view (VIEW_*) of fp32 [B,3,H,W] or uint8 [B,H,W,3] input, inside the stem (split / fp32)"""
        return self.forward(img, None, pixel_mean, pixel_std, out, view=int(view))
