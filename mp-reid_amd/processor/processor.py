"""Drop-in for ``processor/processor.py::do_inference`` (reference processor/processor.py:166-208).

Same loop, log lines and return value; the evaluator and the model are the HIP-backed ones.  The
reference wraps the model in nn.DataParallel when several GPUs are visible (processor/processor.py:178-182:
one process, the batch scattered over the devices).  Here multi-GPU evaluation is one process per GPU
(``python -m torch.distributed.run --nproc-per-node P test.py ...``): under WORLD_SIZE > 1 every rank runs this same
function on its own device, encodes only ITS shard of the validation set -- 1/P of the queries, then its row shard of
the gallery (mpreid.distributed.shard_range) -- and the per-rank ``R1_mAP_eval`` instances meet in ``compute()`` (one
all-gather of the query features, per-shard distance blocks, sharded ranking statistics, host concatenation on rank 0).
Every rank returns the same (Rank-1, Rank-5); rank 0 logs the results.
"""
import logging

import torch

from mpreid import distributed as _D
from utils.metrics import R1_mAP_eval


class _ShardedLoader:
    """The samples [indices] of a validation loader, in order, re-batched with the loader's batch size.

    Fast paths: a loader that knows how to shard itself (``loader.shard(indices)``), and a ``torch.utils.data.DataLoader``
    (a new DataLoader over ``Subset(dataset, indices)`` with the same batch size, workers and collate function, so that
    each rank decodes only its own images).  Any other iterable is filtered: every batch is produced and the samples of
    other ranks are dropped before the model sees them."""

    def __init__(self, loader, indices):
        self.loader, self.indices = loader, list(indices)

    def __iter__(self):
        ld, idx = self.loader, self.indices
        if hasattr(ld, "shard"):
            yield from ld.shard(idx)
            return
        if isinstance(ld, torch.utils.data.DataLoader):
            sub = torch.utils.data.DataLoader(torch.utils.data.Subset(ld.dataset, idx), batch_size=ld.batch_size,
                                              shuffle=False, num_workers=ld.num_workers, collate_fn=ld.collate_fn,
                                              pin_memory=ld.pin_memory)
            yield from sub
            return
        want = set(idx)
        base = 0
        for batch in ld:
            n = len(batch[1])
            keep = [i for i in range(n) if base + i in want]
            base += n
            if keep:
                yield select_samples(batch, keep)


def select_samples(batch, keep):
    """the samples `keep` (positions inside the batch) of a loader batch (img, pids, camids, camids_t, viewids_t, paths)"""
    img, pids, camids, cam_t, view_t, paths = batch
    if len(keep) == len(pids):
        return batch
    if isinstance(img, (list, tuple)):
        img = type(img)(img[i] for i in keep)
    else:
        img = img[torch.as_tensor(keep)]
    sel = torch.as_tensor(keep)
    return (img, tuple(pids[i] for i in keep), tuple(camids[i] for i in keep), cam_t[sel], view_t[sel],
            tuple(paths[i] for i in keep))


def shard_val_loader(val_loader, num_query, n_total=None):
    """this rank's shard of a query-then-gallery validation loader: queries shard_range(num_query, rank, P) followed by
    gallery rows shard_range(n_total - num_query, rank, P); the loader itself when there is one rank"""
    rank, world = _D.rank_world()
    if world == 1:
        return val_loader
    if n_total is None:
        n_total = len(val_loader.dataset) if hasattr(val_loader, "dataset") else getattr(val_loader, "n", None)
    if n_total is None:
        raise ValueError("sharded evaluation needs the size of the validation set (loader.dataset or loader.n)")
    q_lo, q_hi = _D.shard_range(num_query, rank, world)
    g_lo, g_hi = _D.shard_range(n_total - num_query, rank, world)
    return _ShardedLoader(val_loader, list(range(q_lo, q_hi)) + list(range(num_query + g_lo, num_query + g_hi)))


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
    rank, world = _D.rank_world()
    if world > 1:
        logger.info("rank {} of {}: encoding 1/{} of the queries and its gallery shard".format(rank, world, world))
        val_loader = shard_val_loader(val_loader, num_query)
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
    if rank == 0:
        logger.info("Validation Results ")
        logger.info("mAP: {:.1%}".format(mAP))
        for r in [1, 5, 10]:
            logger.info("CMC curve, Rank-{:<3}:{:.1%}".format(r, cmc[r - 1]))
    return cmc[0], cmc[4]
