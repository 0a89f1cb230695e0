"""GPU end-to-end test of the drop-in flow the reference's test.py drives:
make_dataloader(cfg) -> make_model(cfg, ...) -> load_param(path) -> do_inference(cfg, model, val_loader, num_query)
(reference test.py:41-65, processor/processor.py:166-208), checked against the CPU oracle pipeline
(fp32 ViT -> F.normalize -> euclidean / re_ranking -> eval_func)."""
import logging

import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _cfg(nq=24, ng=72, batch=32, neck="before", rerank=False):
    from config import cfg_base
    cfg = cfg_base.clone()
    cfg.defrost()
    cfg.merge_from_list(["DATASETS.SYNTH_QUERY", nq, "DATASETS.SYNTH_GALLERY", ng, "DATASETS.SYNTH_IDS", 12,
                         "TEST.IMS_PER_BATCH", batch, "TEST.NECK_FEAT", neck, "TEST.RE_RANKING", rerank])
    cfg.freeze()
    return cfg


def _oracle_features(model, loader, neck):
    from mpreid import synth
    sd = {k[len("image_encoder."):]: v.cpu().numpy() for k, v in model.state_dict().items()
          if k.startswith("image_encoder.")}
    bn = None
    if neck == "after":
        bn = {n: tuple(getattr(model._modules[n], a).cpu().numpy() for a in ("weight", "bias", "running_mean",
                                                                              "running_var"))
              for n in ("bottleneck", "bottleneck_proj")}
    feats, pids = [], []
    for img, pid, camid, camids, views, paths in loader:
        feats.append(orc.vit_features(sd, model.vit_cfg, img.numpy(), bn=bn, neck_feat=neck))
        pids.extend(pid)
    return np.concatenate(feats), np.asarray(pids)


@pytest.mark.parametrize("neck,rerank", [("before", False), ("after", True)])
def test_do_inference_matches_oracle_pipeline(neck, rerank, caplog):
    from datasets.make_dataloader import make_dataloader
    from model.make_model import make_model
    from processor.processor import do_inference
    from utils.metrics import R1_mAP_eval
    cfg = _cfg(neck=neck, rerank=rerank)
    _, _, val_loader, num_query, num_classes, cam_num, view_num = make_dataloader(cfg)
    model = make_model(cfg, num_class=num_classes, camera_num=cam_num, view_num=view_num)
    if neck == "after":  # non-trivial running statistics so that the BN necks are exercised
        g = torch.Generator().manual_seed(3)
        for n in ("bottleneck", "bottleneck_proj"):
            m = model._modules[n]
            m.running_mean.copy_(0.1 * torch.randn(m.running_mean.shape, generator=g))
            m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=g))
            m.weight.data.copy_(1 + 0.1 * torch.randn(m.weight.shape, generator=g))
            m.bias.data.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
        model._invalidate()
    with caplog.at_level(logging.INFO, logger="transreid.test"):
        r1, r5 = do_inference(cfg, model, val_loader, num_query)
    text = caplog.text
    assert "Validation Results" in text and "mAP: " in text and "CMC curve, Rank-1  :" in text
    # the same thing through the oracle
    feats, pids = _oracle_features(model, val_loader, neck)
    fn = orc.l2_normalize(feats)
    if rerank:
        d_or = orc.re_ranking(fn[:num_query], fn[num_query:], 50, 15, 0.3)
    else:
        d_or = orc.euclidean_distance(fn[:num_query], fn[num_query:])
    # HIP features feed the evaluator again to get at the distance matrix
    ev = R1_mAP_eval(num_query, feat_norm=cfg.TEST.FEAT_NORM, reranking=rerank)
    ev.reset()
    hip_feats = []
    for img, pid, camid, camids, views, paths in val_loader:
        f = model(img.cuda())
        assert f.shape == (img.shape[0], 1280) and f.is_cuda
        hip_feats.append(f.cpu().numpy())
        ev.update((f, pid, camid))
    cmc, mAP, distmat, *_ = ev.compute()
    hip_feats = np.concatenate(hip_feats)
    rel = np.linalg.norm(hip_feats - feats) / np.linalg.norm(feats)
    assert rel < 4e-3, rel
    assert float(cmc[0]) == float(r1) and float(cmc[4]) == float(r5)
    if not rerank:
        # distances between fp16-encoder features vs fp32-oracle features: feature error 4e-3 relative
        assert np.abs(distmat - d_or).max() < 2e-2 * max(1.0, np.abs(d_or).max())


def test_load_param_roundtrip(tmp_path):
    from mpreid import synth
    from model.make_model import make_model
    cfg = _cfg()
    m1 = make_model(cfg, num_class=12, camera_num=6, view_num=1)
    # a "checkpoint" with different weights, DataParallel-style 'module.' prefixes on some keys
    sd = synth.vit_state_dict(synth.VIT_B16, seed=99, std=0.02, ln_jitter=0.05)
    ckpt = {("module.image_encoder." if i % 2 else "image_encoder.") + k: torch.from_numpy(v)
            for i, (k, v) in enumerate(sd.items())}
    path = tmp_path / "ViT-B-16_60.pth"
    torch.save(ckpt, path)
    imgs = torch.from_numpy(synth.synthetic_images(3, 256, 128, seed=8))
    before = m1(imgs).cpu().numpy()
    m1.load_param(str(path))
    after = m1(imgs).cpu().numpy()
    assert np.abs(after - before).max() > 1e-3           # the weights really changed
    want = orc.vit_features(sd, synth.VIT_B16, imgs.numpy())
    rel = np.linalg.norm(after - want) / np.linalg.norm(want)
    assert rel < 4e-3, rel
    # state-dict key layout of the reference's checkpoints
    keys = set(m1.state_dict().keys())
    for k in ("image_encoder.conv1.weight", "image_encoder.transformer.resblocks.11.mlp.c_proj.bias",
              "image_encoder.proj", "bottleneck.running_var", "bottleneck_proj.weight", "classifier.weight"):
        assert k in keys, k


def test_sie_camera_embedding():
    """MODEL.SIE_CAMERA: cv_embed[cam] * SIE_COE is added to the CLS token (model/make_model.py:89-96)"""
    from config import cfg_base
    from mpreid import synth
    from model.make_model import make_model
    cfg = cfg_base.clone()
    cfg.defrost()
    cfg.merge_from_list(["MODEL.SIE_CAMERA", True])
    cfg.freeze()
    m = make_model(cfg, num_class=5, camera_num=6, view_num=1)
    imgs = torch.from_numpy(synth.synthetic_images(4, 256, 128, seed=2))
    cams = torch.tensor([0, 3, 5, 3])
    got = m(imgs.cuda(), cam_label=cams.cuda()).cpu().numpy()
    sd = {k[len("image_encoder."):]: v.cpu().numpy() for k, v in m.state_dict().items()
          if k.startswith("image_encoder.")}
    cv = (cfg.MODEL.SIE_COE * m.cv_embed[cams]).detach().cpu().numpy()
    want = orc.vit_features(sd, m.vit_cfg, imgs.numpy(), cv_emb=cv)
    rel = np.linalg.norm(got - want) / np.linalg.norm(want)
    assert rel < 4e-3, rel
    assert np.abs(got[1] - got[3]).max() > 0  # same camera, different images
