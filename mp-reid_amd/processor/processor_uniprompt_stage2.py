"""Drop-in for the evaluation entry points of the reference's ``processor/processor_uniprompt_stage2.py``:

  do_inference(cfg, model, val_loader, num_query)                 reference :225-266
  do_inference_ttpt_option_a(cfg, model, val_loader, num_query)   reference :530-693  (TTA "option A")

Same loops, log lines and return values.  In option A every QUERY batch is encoded four times (original, W-flip,
pseudo-IR = channel mean, pseudo-RGB = channel 0 replicated), the four features are averaged and L2-normalised
(:598-640); gallery batches are encoded once and normalised (:649-654).  Here the views are produced inside the
patch-gather kernel (no transformed image tensors) and the average + normalise is one kernel
(mpreid_tta_mean_f32).  Quirks kept: whether a batch is "query" is decided by the number of samples processed
BEFORE it (:594), so a batch straddling num_query is augmented as a whole; TEST.FEAT_NORM is a truthy string.

``do_inference_ttpt_clipstyle`` (option B) matches images against text features of the prompt learner and
optimises prompts at test time: text tower, out of scope -> NotImplementedError.
"""
import logging
import time

import torch

from mpreid import ops as _ops
from processor.processor import ENCODE_GROUP, grouped_batches, merge_batches
from utils.metrics import R1_mAP_eval


def do_inference(cfg, model, val_loader, num_query):
    device = "cuda"
    logger = logging.getLogger("transreid.test")
    logger.info("Enter inferencing")

    evaluator = R1_mAP_eval(num_query, max_rank=50, feat_norm=cfg.TEST.FEAT_NORM)
    evaluator.reset()

    model.to(device)
    model.eval()
    img_path_list = []
    for group in grouped_batches(val_loader, ENCODE_GROUP):   # see processor/processor.py: same features, fuller GPU
        with torch.no_grad():
            img, camids, target_view = merge_batches(group, device)
            camids = camids if cfg.MODEL.SIE_CAMERA else None
            target_view = target_view if cfg.MODEL.SIE_VIEW else None
            feat = model(x=img, cam_label=camids, view_label=target_view)
            lo = 0
            for (_, pid, camid, _, _, imgpath) in group:
                evaluator.update((feat[lo:lo + len(pid)], pid, camid))
                img_path_list.extend(imgpath)
                lo += len(pid)

    cmc, mAP, distmat, pids, camids, qf, gf = evaluator.compute()
    logger.info("Validation Results ")
    logger.info("mAP: {:.1%}".format(mAP))
    for r in [1, 5, 10]:
        logger.info("CMC curve, Rank-{:<3}:{:.1%}".format(r, cmc[r - 1]))
    return cmc[0], cmc[4]


def do_inference_ttpt_clipstyle(cfg, model, val_loader, num_query):
    raise NotImplementedError("TTA + TTPT with CLIP-style image-text matching (option B) needs the text tower; "
                              "only option A (image features) is on the accelerated path")


def do_inference_ttpt_option_a(cfg, model, val_loader, num_query):
    device = "cuda"
    logger = logging.getLogger("transreid.test_ttpt_option_a")
    logger.info("Enter inferencing with TTA only (Option A - Image Feature Evaluation)")

    tta_enabled = cfg.TEST.get('TTA_ENABLED', True)
    feat_norm = cfg.TEST.FEAT_NORM
    if tta_enabled:
        logger.info("Test Time Augmentation (TTA) enabled.")
    logger.info("TTPT optimization part is disabled for this run.")

    model.to(device)
    evaluator = R1_mAP_eval(num_query, max_rank=50, feat_norm=cfg.TEST.FEAT_NORM)
    evaluator.reset()
    model.eval()

    logger.info("Starting feature extraction...")
    start_time = time.time()
    processed_samples = 0
    views = (_ops.VIEW_ORIGINAL, _ops.VIEW_FLIP, _ops.VIEW_PSEUDO_IR, _ops.VIEW_PSEUDO_RGB) if tta_enabled else \
        (_ops.VIEW_ORIGINAL,)

    def tagged():
        """loader batches tagged query / gallery by the reference's rule (:594: decided by the number of samples
        processed BEFORE the batch), consecutive batches of one kind grouped up to ENCODE_GROUP images"""
        seen, kind, group, n = 0, None, [], 0
        for batch in val_loader:
            k = seen < num_query
            if group and (k != kind or n >= ENCODE_GROUP):
                yield kind, group
                group, n = [], 0
            kind = k
            group.append(batch)
            n += len(batch[1])
            seen += len(batch[1])
        if group:
            yield kind, group

    for is_query, group in tagged():
        img, cam, vw = merge_batches(group, device)
        cam = cam if cfg.MODEL.SIE_CAMERA else None
        vw = vw if cfg.MODEL.SIE_VIEW else None
        with torch.no_grad():
            if is_query:
                if isinstance(img, (list, tuple)):      # decoded images: resize once, then the views
                    img = _ops.resize_bilinear_u8(img, model.img_hw)
                feats = torch.stack([model(x=img, cam_label=cam, view_label=vw, tta_view=v) for v in views], dim=0)
                feat = _ops.tta_mean(feats, normalize=bool(feat_norm))
            else:
                feat = model(x=img, cam_label=cam, view_label=vw)
                if feat_norm:
                    feat = _ops.l2_normalize(feat)
            lo = 0
            for (_, pid, camid, _, _, _) in group:
                evaluator.update((feat[lo:lo + len(pid)], pid, camid))
                lo += len(pid)
                processed_samples += len(pid)
                if processed_samples % 1000 == 0:
                    logger.info(f"Processed {processed_samples}/{getattr(val_loader, 'n', '?')} samples...")

    logger.info("Feature extraction finished in %.2f seconds." % (time.time() - start_time))

    cmc, mAP = evaluator.compute()[:2]
    return _report_option_a(logger, cmc, mAP, limit=getattr(evaluator, "max_rank", 50))


def _report_option_a(logger, cmc, mAP, limit=50):
    """the result lines of the TTA evaluation (same text as the reference's, processor_uniprompt_stage2.py:676-692: a rank
    that the curve does not reach is reported as such instead of indexed) and its (Rank-1, Rank-5) return value"""
    have = min(len(cmc), limit)
    logger.info("Validation Results (TTPT Option A - Image Features)")
    logger.info("mAP: {:.1%}".format(mAP))
    for r in (1, 5, 10):
        logger.info("CMC curve, Rank-{:<3}:{:.1%}".format(r, cmc[r - 1]) if r <= have else
                    f"Rank-{r} exceeds max_rank ({limit}) or CMC length ({len(cmc)}) ")
    top = [float(cmc[r - 1]) if len(cmc) >= r else 0.0 for r in (1, 5)]
    logger.info("Returning Rank-1: {:.1%}, Rank-5: {:.1%}".format(*top))
    return top[0], top[1]
