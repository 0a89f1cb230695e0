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
import os

import torch

from mpreid import distributed as _D
from mpreid.pipeline import EncodePipeline
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


#: images encoded per model call when the model does not say (``model.encode_group``).  The reference encodes one
#: loader batch (TEST.IMS_PER_BATCH = 64 in every shipped YAML) at a time; the persistent GEMMs want ~65 000 token rows
#: to fill 256 CUs without a ragged last round (22 k images/s at 64 per call, 36 k at 512), and a feature row does not
#: depend on which images share its batch (tests/test_gpu_vit.py: bit for bit), so consecutive loader batches are
#: encoded together -- cut at group boundaries, not batch boundaries -- and handed to the evaluator one loader batch at
#: a time, exactly as before.
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
    # TEST.DISTANCE_MODE / TEST.RERANK_ALGO (not reference keys): arithmetic of the distance matrix -- 'exact' (default: the
    # bit-parity fp32 chain), 'split3' (fp16 matrix cores, |err| <= 1e-6), 'f16' (one pass, ~1e-4) -- and of the re-ranking
    # ('exact' | 'split3': blend-term distance rows from the fp16 matrix cores, ranks identical, outputs within 1e-6)
    from mpreid import ops as _ops
    evaluator.distance_mode = {"exact": _ops.GEMM_F32_EXACT, "split3": _ops.GEMM_F16_SPLIT3, "f16": _ops.GEMM_F16_FAST}[
        str(getattr(cfg.TEST, "DISTANCE_MODE", "exact"))]
    evaluator.rerank_algo = {"exact": _ops.RERANK_AUTO, "split3": _ops.RERANK_SPARSE_SPLIT3}[
        str(getattr(cfg.TEST, "RERANK_ALGO", "exact"))]
    evaluator.reset()

    # the reference's test.py never initialises a process group (test.py:39,65): under a launcher (WORLD_SIZE > 1) this
    # call does, and binds the rank to cuda:LOCAL_RANK, before anything touches a device (raises when the ranks of the
    # node cannot each see their own device -- MODEL.DEVICE_ID must list them: INTEGRATION.md section C)
    rank, world = _D.ensure_group_from_env(logger)
    model.to(device)
    model.eval()
    img_path_list = []
    if world > 1:
        logger.info("rank {} of {}: encoding 1/{} of the queries and its gallery shard".format(rank, world, world))
        val_loader = shard_val_loader(val_loader, num_query)
    # the reference's loop body (`img = img.to(device); feat = model(img, ...); evaluator.update(...)`, :187-198) as a
    # pipeline: a stager thread drains the loader into pinned group buffers and uploads them on a copy stream while
    # earlier groups are encoded on two alternating streams (mpreid/pipeline.py); the evaluator sees the loader's own
    # batches, in order.  MPREID_PIPELINE="stage=pinned,streams=1,slots=4" overrides the defaults (measurements).
    opts = dict(kv.split("=", 1) for kv in os.environ.get("MPREID_PIPELINE", "").split(",") if "=" in kv)
    pipe = EncodePipeline(model, group=int(opts.get("group", getattr(model, "encode_group", ENCODE_GROUP))),
                          sie_camera=bool(cfg.MODEL.SIE_CAMERA), sie_view=bool(cfg.MODEL.SIE_VIEW),
                          streams=int(opts.get("streams", 2)), slots=int(opts.get("slots", 3)),
                          stage=opts.get("stage", "direct"))
    for feat, (_, pid, camid, _, _, imgpath) in pipe.run(val_loader):
        evaluator.update((feat, pid, camid))
        img_path_list.extend(imgpath)
    do_inference.last_pipeline_stats = pipe.stats

    cmc, mAP, _, _, _, _, _ = evaluator.compute()
    do_inference.last_evaluator = evaluator   # (measurements: bench.py times compute() again on the same features)
    if rank == 0:
        logger.info("Validation Results ")
        logger.info("mAP: {:.1%}".format(mAP))
        for r in [1, 5, 10]:
            logger.info("CMC curve, Rank-{:<3}:{:.1%}".format(r, cmc[r - 1]))
    return cmc[0], cmc[4]
