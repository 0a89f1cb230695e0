"""Drop-in for ``processor/processor.py::do_inference`` (reference processor/processor.py:166-208).

Same loop, log lines and return value; the evaluator and the model are the HIP-backed ones.  The
reference wraps the model in nn.DataParallel when several GPUs are visible; here multi-GPU
evaluation is one process per GPU (mpreid/distributed.py, used by bench.py) and this function
always drives the current device.
"""
import logging

import torch

from utils.metrics import R1_mAP_eval


#: images encoded per model call.  The reference encodes one loader batch (TEST.IMS_PER_BATCH = 64 in every shipped
#: YAML) at a time; the persistent GEMMs want ~65 000 token rows to fill 256 CUs without a ragged last round
#: (22 k images/s at 64 per call, 36 k at 512), and a feature row does not depend on which images share its batch
#: (tests/test_gpu_vit.py: bit for bit), so consecutive loader batches are encoded together and handed to the
#: evaluator one loader batch at a time, exactly as before.
ENCODE_GROUP = 512


def grouped_batches(val_loader, target):
    """consecutive loader batches whose image count reaches `target` (the last group may be smaller)"""
    group, n = [], 0
    for batch in val_loader:
        group.append(batch)
        n += len(batch[1])
        if n >= target:
            yield group
            group, n = [], 0
    if group:
        yield group


def merge_batches(group, device):
    """(images, camids tensor, viewids tensor) of a group of loader batches, on `device`"""
    imgs = [b[0] for b in group]
    if isinstance(imgs[0], (list, tuple)):           # decoded uint8 images: stay a flat list (RawImageBatch)
        img = type(imgs[0])(x for b in imgs for x in b)
    else:
        img = (imgs[0] if len(imgs) == 1 else torch.cat(imgs, dim=0)).to(device)
    camids = torch.cat([b[3] for b in group]).to(device)
    views = torch.cat([b[4] for b in group]).to(device)
    return img, camids, views


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
    for group in grouped_batches(val_loader, ENCODE_GROUP):
        with torch.no_grad():
            img, camids, target_view = merge_batches(group, device)
            camids = camids if cfg.MODEL.SIE_CAMERA else None
            target_view = target_view if cfg.MODEL.SIE_VIEW else None
            feat = model(img, cam_label=camids, view_label=target_view)
            lo = 0
            for (_, pid, camid, _, _, imgpath) in group:      # the evaluator sees the loader's own batches
                evaluator.update((feat[lo:lo + len(pid)], pid, camid))
                img_path_list.extend(imgpath)
                lo += len(pid)

    cmc, mAP, _, _, _, _, _ = evaluator.compute()
    logger.info("Validation Results ")
    logger.info("mAP: {:.1%}".format(mAP))
    for r in [1, 5, 10]:
        logger.info("CMC curve, Rank-{:<3}:{:.1%}".format(r, cmc[r - 1]))
    return cmc[0], cmc[4]
