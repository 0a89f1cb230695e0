"""Drop-in for ``processor/processor.py::do_inference`` (reference processor/processor.py:166-208).

Same loop, log lines and return value; the evaluator and the model are the HIP-backed ones.  The
reference wraps the model in nn.DataParallel when several GPUs are visible; here multi-GPU
evaluation is one process per GPU (mpreid/distributed.py, used by bench.py) and this function
always drives the current device.
"""
import logging

import torch

from utils.metrics import R1_mAP_eval


def do_inference(cfg, model, val_loader, num_query):
    device = "cuda"
    logger = logging.getLogger("transreid.test")
    logger.info("Enter inferencing")

    reranking = bool(getattr(cfg.TEST, "RE_RANKING", False))  # upstream defines the key but never reads it
    evaluator = R1_mAP_eval(num_query, max_rank=50, feat_norm=cfg.TEST.FEAT_NORM, reranking=reranking)
    evaluator.reset()

    model.to(device)
    model.eval()
    img_path_list = []
    for n_iter, (img, pid, camid, camids, target_view, imgpath) in enumerate(val_loader):
        with torch.no_grad():
            img = img.to(device)
            camids = camids.to(device) if cfg.MODEL.SIE_CAMERA else None
            target_view = target_view.to(device) if cfg.MODEL.SIE_VIEW else None
            feat = model(img, cam_label=camids, view_label=target_view)
            evaluator.update((feat, pid, camid))
            img_path_list.extend(imgpath)

    cmc, mAP, _, _, _, _, _ = evaluator.compute()
    logger.info("Validation Results ")
    logger.info("mAP: {:.1%}".format(mAP))
    for r in [1, 5, 10]:
        logger.info("CMC curve, Rank-{:<3}:{:.1%}".format(r, cmc[r - 1]))
    return cmc[0], cmc[4]
