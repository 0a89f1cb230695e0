"""Drop-in for the reference's ``model/make_model.py`` (eval path only): the CLIP ViT-B/16 image
encoder + feature head of ``build_transformer``, executed by the HIP kernels of libmpreid_hip.so.

Kept from the reference (model/make_model.py:81-133): ``make_model(cfg, num_class, camera_num, view_num)``
returns an object with ``.load_param(path)``, ``.eval()``, ``.to(device)``, ``.state_dict()`` (same key
layout as the reference's checkpoints: ``image_encoder.*``, ``bottleneck*``, ``classifier*``, ``cv_embed``)
and ``__call__(x, label=None, cam_label=None, view_label=None) -> Tensor[B, 1280]`` (3072 for RN50).

Not kept on purpose: the constructor does not download the pretrained CLIP archive (the reference
does, even at test time, model/make_model.py:137-139); weights are seeded random until
``load_param`` is called.  MODEL.NAME 'RN50' selects the CLIP ModifiedResNet tower (2048 + 1024 = 3072-d feature,
model/make_model.py:40-42, 82-86).  Training-mode forward and the text tower are out of scope.
"""
import numpy as np
import torch
import torch.nn as nn

from mpreid import ops as _ops
from mpreid import synth as _synth


def _put(root: nn.Module, dotted: str, tensor: torch.Tensor):
    """register `tensor` as a parameter at a dotted state-dict path, creating containers on the way"""
    parts = dotted.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, nn.Module())
        mod = mod._modules[p]
    mod.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=False))


class build_transformer(nn.Module):
    def __init__(self, num_classes, camera_num, view_num, cfg):
        super().__init__()
        self.model_name = cfg.MODEL.NAME
        if self.model_name not in ('ViT-B-16', 'RN50'):
            raise NotImplementedError(f"MODEL.NAME {self.model_name!r}: the reference knows 'ViT-B-16' and 'RN50'")
        self.neck_feat = cfg.TEST.NECK_FEAT
        # MODEL.ENCODER_PRECISION (not a reference key; the reference encodes in fp32, processor/processor.py:187-198):
        # 'split' (default) = fp16 operand pairs hi + lo on the fp16 matrix cores, fp32-grade features, meets the 1e-4
        # mAP bound; 'fp16' = single fp16 operands (fastest, ~4e-4 feature error, misses the bound on hard data);
        # 'fp32' = the all-fp32 mode on the exact fp32 matrix instruction (mpreid_vit_forward_f32)
        self.precision = str(getattr(cfg.MODEL, "ENCODER_PRECISION", "split"))
        self.in_planes, self.in_planes_proj = (768, 512) if self.model_name == 'ViT-B-16' else (2048, 1024)
        self.num_classes, self.camera_num, self.view_num = num_classes, camera_num, view_num
        self.sie_coe = cfg.MODEL.SIE_COE
        stride = cfg.MODEL.STRIDE_SIZE[0]
        # the positional grid follows INPUT.SIZE_TRAIN, like upstream (it must equal SIZE_TEST)
        self.h_resolution = int((cfg.INPUT.SIZE_TRAIN[0] - 16) // cfg.MODEL.STRIDE_SIZE[0] + 1)
        self.w_resolution = int((cfg.INPUT.SIZE_TRAIN[1] - 16) // cfg.MODEL.STRIDE_SIZE[1] + 1)
        self.img_hw = (int(cfg.INPUT.SIZE_TEST[0]), int(cfg.INPUT.SIZE_TEST[1]))
        self.pixel_mean = tuple(float(v) for v in cfg.INPUT.PIXEL_MEAN)
        self.pixel_std = tuple(float(v) for v in cfg.INPUT.PIXEL_STD)
        self.vit_cfg = dict(h_res=self.h_resolution, w_res=self.w_resolution, patch=16, stride=stride, width=768,
                            layers=12, heads=12, out_dim=512)
        seed = int(getattr(cfg.MODEL, "INIT_SEED", 7))
        if self.model_name == 'RN50':
            # model/clip/model.py:509-516: heads = width * 32 // 64, attention-pool grid = h_resolution x w_resolution
            self.rn_cfg = dict(_synth.RN50, h_res=self.h_resolution, w_res=self.w_resolution)
            for k, v in _synth.rn50_state_dict(self.rn_cfg, seed=seed).items():
                path = "image_encoder." + k
                if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
                    parts = path.split(".")
                    mod = self
                    for p_ in parts[:-1]:
                        if p_ not in mod._modules:
                            mod.add_module(p_, nn.Module())
                        mod = mod._modules[p_]
                    mod.register_buffer(parts[-1], torch.from_numpy(np.asarray(v)))
                else:
                    _put(self, path, torch.from_numpy(v))
        else:
            for k, v in _synth.vit_state_dict(self.vit_cfg, seed=seed).items():
                _put(self, "image_encoder." + k, torch.from_numpy(v))
        g = torch.Generator().manual_seed(seed)
        _put(self, "classifier.weight", torch.randn(num_classes, self.in_planes, generator=g) * 0.001)
        _put(self, "classifier_proj.weight", torch.randn(num_classes, self.in_planes_proj, generator=g) * 0.001)
        for name, n in (("bottleneck", self.in_planes), ("bottleneck_proj", self.in_planes_proj)):
            _put(self, name + ".weight", torch.ones(n))
            _put(self, name + ".bias", torch.zeros(n))
            bn = self._modules[name]
            bn.register_buffer("running_mean", torch.zeros(n))
            bn.register_buffer("running_var", torch.ones(n))
            bn.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.sie_camera, self.sie_view = bool(cfg.MODEL.SIE_CAMERA), bool(cfg.MODEL.SIE_VIEW)
        if self.sie_camera or self.sie_view:
            rows = camera_num * view_num if (self.sie_camera and self.sie_view) else (
                camera_num if self.sie_camera else view_num)
            cv = torch.empty(rows, self.in_planes)
            nn.init.trunc_normal_(cv, std=.02, generator=g)
            self.cv_embed = nn.Parameter(cv, requires_grad=False)
        self._encoder = None
        self.eval()

    # -- device encoder is (re)built lazily from the current parameters --------------------------
    def _invalidate(self):
        self._encoder = None

    def _build_encoder(self, neck_after, ws_tag=None):
        sd = {k: v for k, v in self.state_dict().items() if k.startswith("image_encoder.")}
        bn = None
        if neck_after:
            bn = {n: (self._modules[n].weight, self._modules[n].bias, self._modules[n].running_mean,
                      self._modules[n].running_var) for n in ("bottleneck", "bottleneck_proj")}
        kw = {} if ws_tag is None else {"ws_tag": ws_tag}
        if self.model_name == 'RN50':
            # RN50 modes: 'split' (default: fp32 activations, layer1-4 and the attention pool's k / v projections over fp16
            # pairs on the fp16 matrix cores -- fp32-grade features), 'fp32' (everything on the exact fp32 matrix instruction)
            # and 'fp16' (fp16 activations: the throughput tower, feature error 2.6e-3, misses the 1e-4 mAP bound)
            assert self.precision in ("split", "fp32", "fp16"), self.precision
            return _ops.Rn50Encoder(self.rn_cfg, sd, self.img_hw, neck_after=neck_after, bn=bn, precision=self.precision, **kw)
        return _ops.VitEncoder(self.vit_cfg, sd, self.img_hw, neck_after=neck_after, bn=bn, precision=self.precision, **kw)

    @property
    def encode_group(self):
        """images per encoder call that fill the chip without a ragged last round: the persistent GEMMs walk 256-row tiles on
        256 CUs, so the largest batch whose token rows fit 256 * 256 (508 images of 129 tokens); 256 images for RN50
        (processor.do_inference groups the loader's batches to this size)"""
        if self.model_name == 'RN50':
            return 256
        tokens = self.h_resolution * self.w_resolution + 1
        return max(64, 65536 // tokens)

    def _get_encoder(self):
        if self._encoder is None:
            self._encoder = self._build_encoder(self.neck_feat == 'after')
        return self._encoder

    def forward(self, x, label=None, cam_label=None, view_label=None):
        if self.training:
            raise NotImplementedError("training-mode forward is out of scope; call .eval()")
        return self._encode(x, self._sie(cam_label, view_label), 0)

    def _sie(self, cam_label, view_label):
        cv_embed = None
        if cam_label is not None or view_label is not None:
            # SIE: index = cam * view_num + view | cam | view (reference model/make_model.py:89-96); the table may
            # live on the CPU (the module was not .to("cuda")-ed) while the labels are on the GPU
            if cam_label is not None and view_label is not None:
                idx = cam_label * self.view_num + view_label
            else:
                idx = cam_label if cam_label is not None else view_label
            cv_embed = self.sie_coe * self.cv_embed[idx.to(self.cv_embed.device)]
        return cv_embed

    def _encode(self, x, cv_embed, view):
        """x: fp32 [B,3,H,W] (val_transforms already applied), uint8 [B,H,W,3] (after Resize) or a RawImageBatch /
        list of decoded uint8 [h,w,3] images (Resize, ToTensor and Normalize run on the GPU with INPUT.PIXEL_MEAN /
        PIXEL_STD).  view: a test-time-augmentation view id (mpreid.ops.VIEW_*)."""
        enc = self._get_encoder()
        if isinstance(x, (list, tuple, _ops.PackedRawImages)):
            x = _ops.resize_bilinear_u8(x, self.img_hw)
        if self.model_name == 'RN50':
            if view != 0 and self.precision in ("split", "fp32"):
                # (round 5) the view transform happens inside the stem's first convolution (mpreid_rn50_forward_*_view)
                return enc.forward_view(x, view, None, self.pixel_mean, self.pixel_std)
            if view != 0:
                # the fp16 throughput tower: the view tensor is materialised on the device the way the reference does it
                # (processor/processor_uniprompt_stage2.py:605-633)
                if x.dtype == torch.uint8:
                    mean = torch.tensor(self.pixel_mean, device=x.device)[None, :, None, None]
                    std = torch.tensor(self.pixel_std, device=x.device)[None, :, None, None]
                    x = (x.permute(0, 3, 1, 2).to(torch.float32).div(255) - mean) / std
                x = x.to(enc.device)
                x = (torch.flip(x, [3]) if view == 1 else
                     x.mean(dim=1, keepdim=True).repeat(1, 3, 1, 1) if view == 2 else x[:, 0:1].repeat(1, 3, 1, 1))
                x = x.contiguous()
            return enc(x, None, self.pixel_mean, self.pixel_std)   # no SIE embedding in the reference's RN50 branch
        if view == 0 and x.dtype != torch.uint8:
            return enc(x, cv_embed)
        return enc.forward_view(x, view, cv_embed, self.pixel_mean, self.pixel_std)

    def load_param(self, trained_path):
        param_dict = torch.load(trained_path, map_location="cpu")
        own = self.state_dict()
        for name in param_dict:
            own[name.replace('module.', '')].copy_(param_dict[name])
        self._invalidate()
        print('Loading pretrained model from {}'.format(trained_path))

    def load_param_finetune(self, model_path):
        self.load_param(model_path)

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._invalidate()
        return r


def make_model(cfg, num_class, camera_num, view_num):
    return build_transformer(num_class, camera_num, view_num, cfg)
