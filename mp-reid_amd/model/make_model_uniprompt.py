"""Drop-in for the reference's ``model/make_model_uniprompt.py`` — evaluation branches only.

The Uni-Prompt ``build_transformer`` computes its evaluation feature exactly like the base model (reference
model/make_model_uniprompt.py:209-238: image encoder with the SIE embedding, CLS of the last block and of its
projection, eval BatchNorm, concat per ``TEST.NECK_FEAT``), so the HIP encoder of ``model/make_model.py`` serves it.
Kept from the reference signature (:159): ``forward(x=None, label=None, get_image=False, get_text=False,
cam_label=None, view_label=None, ..., get_image_vp=False)``:

  default          -> Tensor[B, 1280]           (:203-238, what do_inference / the TTA evaluation call)
  get_image=True   -> projected CLS [B, 512]    (:172-177)
  get_image_vp     -> projected CLS + visual_prompt (:178-186)

The text tower (``get_text`` / ``get_raw_text``: PromptLearner + CLIP text transformer), the MLP feature fusion
(``get_image_update``) and the multi-token variant (``get_more_image``) belong to training / TTPT and are out of
scope: they raise NotImplementedError.  ``load_param`` accepts Uni-Prompt checkpoints and skips the keys of those
modules (the reference would copy them into modules this build does not have).

``tta_view`` (not in the reference) selects a test-time-augmentation view computed inside the patch gather; the
reference materialises the view tensors instead (processor/processor_uniprompt_stage2.py:605-633) — passing such a
tensor as ``x`` works as well and gives the same bits.
"""
import torch
import torch.nn as nn

from . import make_model as _base

_SKIPPED_PREFIXES = ("prompt_learner.", "text_encoder.", "image_fusion_net.")


class build_transformer(_base.build_transformer):
    def __init__(self, num_classes, camera_num, view_num, cfg):
        super().__init__(num_classes, camera_num, view_num, cfg)
        self.prompt_dim = 512   # hard-coded in the reference (:89), whatever the backbone
        g = torch.Generator().manual_seed(int(getattr(cfg.MODEL, "INIT_SEED", 7)) + 1)
        self.visual_prompt = nn.Parameter(torch.randn(1, 1, self.prompt_dim, generator=g) * 0.02, requires_grad=False)
        self._proj_encoder = None

    def _invalidate(self):
        super()._invalidate()
        self._proj_encoder = None

    def _raw_features(self, x, view):
        """cat(x12[:,0], xproj[:,0]) before any BatchNorm, whatever TEST.NECK_FEAT says"""
        if self.neck_feat != 'after':
            return self._encode(x, None, view)
        if self._proj_encoder is None:
            self._proj_encoder = self._build_encoder(False, ws_tag="enc_proj")
        keep, self._encoder = self._encoder, self._proj_encoder
        try:
            return self._encode(x, None, view)
        finally:
            self._encoder = keep

    def forward(self, x=None, label=None, get_image=False, get_text=False, cam_label=None, view_label=None,
                image_feature=None, get_raw_text=False, view=None, get_image_update=False, text_feature=None,
                get_more_image=False, exp_setting=None, get_image_vp=False, tta_view=0):
        if self.training:
            raise NotImplementedError("training-mode forward is out of scope; call .eval()")
        if get_text or get_raw_text or get_image_update or get_more_image:
            raise NotImplementedError("text tower / feature fusion / multi-token outputs belong to Uni-Prompt "
                                      "training and TTPT, outside the accelerated evaluation path (SURVEY.md §8f)")
        if get_image or get_image_vp:
            proj = self._raw_features(x, tta_view)[:, self.in_planes:]
            return proj + self.visual_prompt[0].to(proj.device) if get_image_vp else proj
        return self._encode(x, self._sie(cam_label, view_label), tta_view)

    def load_param(self, trained_path):
        param_dict = torch.load(trained_path, map_location="cpu")
        own = self.state_dict()
        skipped = 0
        for name in param_dict:
            key = name.replace('module.', '')
            if key.startswith(_SKIPPED_PREFIXES):
                skipped += 1
                continue
            own[key].copy_(param_dict[name])
        self._invalidate()
        print('Loading pretrained model from {}'.format(trained_path))
        if skipped:
            print('  ({} text-tower / fusion tensors are not used at evaluation and were skipped)'.format(skipped))

    def load_param_finetune(self, model_path):
        self.load_param(model_path)


def make_model(cfg, num_class, camera_num, view_num):
    return build_transformer(num_class, camera_num, view_num, cfg)
