"""Drop-in for the reference's ``utils/metrics.py``: distance matrices, CMC/mAP and the
``R1_mAP_eval`` accumulator, with the heavy steps on the MI355X (libmpreid_hip.so).

Signatures kept (reference utils/metrics.py): euclidean_distance(qf, gf) :7, cosine_similarity(qf, gf) :15,
eval_func(distmat, q_pids, g_pids, q_camids, g_camids, max_rank=50) :28,
R1_mAP_eval(num_query, max_rank=50, feat_norm=True, reranking=False) with reset/update/compute :91-134.

Differences that are deliberate and invisible to callers:
  * update() keeps a device-resident copy of each feature batch instead of ``feat.cpu()``; the
    whole of compute(), including the ranking behind CMC/mAP (eval_func_device), runs on the GPU.
    ``eval_func`` itself (numpy in, numpy out) stays a host function with the reference's signature.
  * ties in eval_func's ranking are broken by ascending gallery index (the reference's
    np.argsort is unstable, so it has no defined order on ties).
"""
import numpy as np
import torch

from mpreid import ops as _ops
from utils.reranking import re_ranking, re_ranking_device


def _as_tensor(x):
    return torch.from_numpy(x) if isinstance(x, np.ndarray) else x


def euclidean_distance(qf, gf):
    """Squared Euclidean distance |q|^2 + |g|^2 - 2 q.g  ->  np.float32 [m, n]."""
    return _ops.euclidean_distance(_as_tensor(qf), _as_tensor(gf)).cpu().numpy()


def cosine_similarity(qf, gf):
    """arccos of the clipped cosine  ->  np.float32 [m, n]."""
    return _ops.cosine_similarity(_as_tensor(qf), _as_tensor(gf)).cpu().numpy()


def eval_func(distmat, q_pids, g_pids, q_camids, g_camids, max_rank=50):
    """Market-1501 style CMC and mAP exactly as the reference computes them: gallery samples that
    share pid AND camera with the query are NOT removed (that filter is disabled upstream), so the
    camera ids are accepted and unused.  Host-side, vectorised over queries."""
    distmat = np.asarray(distmat)
    q_pids = np.asarray(q_pids)
    g_pids = np.asarray(g_pids)
    num_q, num_g = distmat.shape
    if num_g < max_rank:
        max_rank = num_g
        print("Note: number of gallery samples is quite small, got {}".format(num_g))
    order = np.argsort(distmat, axis=1, kind="stable")
    hit = (g_pids[order] == q_pids[:, None])
    valid = hit.any(axis=1)
    num_valid_q = float(valid.sum())
    assert num_valid_q > 0, "Error: all query identities do not appear in gallery"
    hit = hit[valid]
    running = np.cumsum(hit, axis=1)
    # CMC: 1 from the first correct match on
    cmc_rows = (running[:, :max_rank] > 0).astype(np.float32)
    all_cmc = cmc_rows.sum(0) / num_valid_q
    # AP = mean over relevant ranks k of (hits up to k) / k
    ranks = np.arange(1, num_g + 1) * 1.0
    prec_at_hit = (running / ranks) * hit
    all_AP = prec_at_hit.sum(axis=1) / hit.sum(axis=1)
    mAP = np.mean(all_AP)
    return all_cmc, mAP


_warned_host_ranking = False


def _eval_rows_device(dist, q_pids, g_pids, max_rank, after_launch=None):
    """Ranking statistics of the query ROWS in `dist` (device tensor [rows, ng] fp32): (cmc hit counts [max_rank] float32
    summed over the valid rows, AP of every valid row in row order (float64), number of valid rows).  Sums of 0/1 values
    are exact in float32, so hit counts of row shards add up to the unsharded counts bit for bit.
    `after_launch()` is called once the ranking kernel is queued and before the host waits for it (compute() starts the
    matrix's D2H copy there, ordered BEHIND the kernel)."""
    import ctypes as C
    from mpreid import _lib
    dev = _lib.require_gpu()
    L = _lib.load()
    dist = dist.detach()
    assert dist.is_cuda and dist.dtype == torch.float32 and dist.dim() == 2 and (dist.stride(1) == 1 or dist.shape[0] == 0)
    num_q, num_g = dist.shape
    q_pids = np.ascontiguousarray(q_pids, dtype=np.int64)
    g_pids = np.ascontiguousarray(g_pids, dtype=np.int64)
    if num_q == 0:
        if after_launch is not None:
            after_launch()
        return np.zeros(max_rank, np.float32), np.zeros(0, np.float64), 0
    rcap = int(min(max(np.unique(g_pids, return_counts=True)[1].max(), 1), 8192))   # (the kernel's LDS limit, include/mpreid.h)
    qp, gp = torch.from_numpy(q_pids).to(dev), torch.from_numpy(g_pids).to(dev)
    pos = torch.empty((num_q, rcap), dtype=torch.int32, device=dev)
    cnt = torch.empty(num_q, dtype=torch.int32, device=dev)
    _lib.check(L.mpreid_eval_rank_positions(C.c_void_p(dist.data_ptr()), dist.stride(0), num_q, num_g,
                                            C.c_void_p(qp.data_ptr()), C.c_void_p(gp.data_ptr()), rcap,
                                            C.c_void_p(pos.data_ptr()), C.c_void_p(cnt.data_ptr()), _lib.stream_ptr()),
               "mpreid_eval_rank_positions")
    if after_launch is not None:
        after_launch()
    pos, cnt = pos.cpu().numpy().astype(np.int64), cnt.cpu().numpy().astype(np.int64)
    over = np.nonzero(cnt < 0)[0]          # queries with more relevant items than the kernel handles: host ranking
    if over.size:
        global _warned_host_ranking
        if not _warned_host_ranking:
            _warned_host_ranking = True
            import logging
            logging.getLogger("transreid.test").warning(
                "eval_func: %d of %d queries have more than %d relevant gallery items; their rows are ranked on the host "
                "(np.argsort per row, same result)", over.size, num_q, rcap)
        pos = np.concatenate([pos, np.full((num_q, 0), -1, np.int64)], axis=1)
        rows = dist[torch.from_numpy(over).to(dev)].cpu().numpy()
        wide = max(int((g_pids[None, :] == q_pids[over, None]).sum(1).max()), pos.shape[1])
        pos = np.pad(pos, ((0, 0), (0, wide - pos.shape[1])), constant_values=-1)
        for r, qi in enumerate(over):
            order = np.argsort(rows[r], kind="stable")
            p = np.nonzero(g_pids[order] == q_pids[qi])[0]
            pos[qi, :] = -1
            pos[qi, :p.size] = p
            cnt[qi] = p.size
    valid = cnt > 0
    num_valid = int(valid.sum())
    if num_valid == 0:
        return np.zeros(max_rank, np.float32), np.zeros(0, np.float64), 0
    pos, cnt = pos[valid], cnt[valid]
    first = pos[:, 0]
    cmc_rows = (np.arange(max_rank)[None, :] >= first[:, None]).astype(np.float32)
    t = np.arange(1, pos.shape[1] + 1, dtype=np.float64)[None, :]
    terms = np.where(pos >= 0, t / np.maximum(pos + 1.0, 1.0), 0.0)
    return cmc_rows.sum(0), terms.sum(axis=1) / cnt, num_valid


def eval_func_device(dist, q_pids, g_pids, q_camids=None, g_camids=None, max_rank=50, after_launch=None):
    """eval_func with the ranking done on the GPU (dist: device tensor [nq, ng] fp32, left on the device).

    Per query the kernel returns the positions of the relevant gallery items in the ascending (distance, index)
    order of the row — what the reference reads off np.argsort — in one pass over the row; CMC and AP are
    finished here in float64.  CMC is identical to eval_func's; AP sums the same terms (hits up to k) / k in a
    different order than numpy's pairwise reduction over the dense row, i.e. |delta mAP| ~ 1e-16.

    Under a process group (one rank per GPU) `dist` may be this rank's ROW block and q_pids its rows' pids: see
    eval_func_sharded."""
    num_g = dist.shape[1]
    if num_g < max_rank:
        max_rank = num_g
        print("Note: number of gallery samples is quite small, got {}".format(num_g))
    hits, ap, num_valid = _eval_rows_device(dist, q_pids, g_pids, max_rank, after_launch)
    assert num_valid > 0, "Error: all query identities do not appear in gallery"
    return hits / float(num_valid), np.mean(ap)


def eval_func_sharded(dist_rows, q_pids_local, g_pids, max_rank=50):
    """eval_func over query rows sharded across the ranks of the default process group (SURVEY.md section 8e, row
    `eval_func`): every rank ranks its own rows [q_lo, q_hi) on its GPU; the hit counts (exact small integers) are
    summed and the per-query AP lists are concatenated in rank = query order, so every rank ends up with the cmc / mAP
    of the unsharded call BIT FOR BIT (integer sums; np.mean over the same float64 list in the same order)."""
    import torch.distributed  # noqa: F401  (ReduceOp)
    from mpreid import distributed as D
    rank, world = D.rank_world()
    num_g = dist_rows.shape[1]
    if num_g < max_rank:
        max_rank = num_g
        if rank == 0:
            print("Note: number of gallery samples is quite small, got {}".format(num_g))
    hits, ap, num_valid = _eval_rows_device(dist_rows, q_pids_local, g_pids, max_rank)
    if D.sharded_active():
        staged = D._pg().get_backend() == "gloo"
        dev = "cpu" if staged else dist_rows.device
        # one small all-gather: [hit counts (max_rank) | number of valid rows | AP of the valid rows, NaN padded]
        cap = torch.tensor([ap.size], dtype=torch.int64, device=dev)
        D._pg().all_reduce(cap, op=torch.distributed.ReduceOp.MAX)
        cap = int(cap.item())
        msg = np.full(max_rank + 1 + cap, np.nan, np.float64)
        msg[:max_rank] = hits
        msg[max_rank] = num_valid
        msg[max_rank + 1: max_rank + 1 + ap.size] = ap
        parts = [torch.empty(msg.size, dtype=torch.float64, device=dev) for _ in range(world)]
        D._pg().all_gather(parts, torch.from_numpy(msg).to(dev))
        parts = [p.cpu().numpy() for p in parts]
        hits = np.sum([p[:max_rank] for p in parts], axis=0).astype(np.float32)   # exact: integers < 2^24
        num_valid = int(sum(p[max_rank] for p in parts))
        ap = np.concatenate([p[max_rank + 1: max_rank + 1 + int(p[max_rank])] for p in parts])
    assert num_valid > 0, "Error: all query identities do not appear in gallery"
    return hits / float(num_valid), np.mean(ap)


def _check_finite(feats, collective=False):
    """An encoder whose fp16 operand halves overflowed (|activation| > 65 504 in the 'split' / 'fp16' precision modes:
    include/mpreid.h, mpreid_vit_forward) hands over NaN / inf feature rows; ranking them would print a plausible-looking
    mAP.  Refuse loudly instead (one reduction over [N, D]; compute() synchronises anyway)."""
    import os
    if os.environ.get("MPREID_CHECK_FINITE", "1") == "0":   # opt-out: the reference's behaviour (it ranks whatever it is given)
        return
    bad = int((~torch.isfinite(feats).all(dim=1)).sum()) if feats.numel() else 0
    if collective:   # every rank must take the same branch: a rank that raised alone would leave the others in a collective
        from mpreid import distributed as D
        t = torch.tensor([bad], dtype=torch.int64, device="cpu" if D._pg().get_backend() == "gloo" else feats.device)
        D._pg().all_reduce(t)
        bad = int(t.item())
    if bad:
        raise RuntimeError(f"R1_mAP_eval.compute(): {bad} feature rows are non-finite -- the encoder's "
                           "fp16 operands overflowed (or the model produced NaN); use MODEL.ENCODER_PRECISION fp32 for "
                           "this checkpoint / input range (MPREID_CHECK_FINITE=0 restores the reference's behaviour: "
                           "no check, the rows are ranked as they are)")


_d2h_streams = {}


def _to_host_async(tensors):
    """Start the D2H copies of device tensors into fresh page-locked host tensors on a side stream (ordered after the
    current stream's work so far); returns (host tensors, event to synchronise before reading them).  The host tensors
    come from torch's caching pinned allocator: the caller owns them like any CPU tensor, and repeated evaluations reuse
    the pages.  (A pageable ``.cpu()`` of the 214 MB Market-1501 matrix is staged through a bounce buffer by the runtime and
    blocks the host for its whole duration; this way the copies run while the ranking kernel does.)"""
    dev = tensors[0].device
    side = _d2h_streams.get(dev)
    if side is None:
        side = _d2h_streams[dev] = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    hosts = []
    import os
    cap = int(os.environ.get("MPREID_PINNED_CAP_MB", "1024")) << 20
    with torch.cuda.stream(side):
        for t in tensors:
            # page-locked up to the cap (Market-1501's matrix is 214 MB); a multi-GB matrix (MSMT17: 3.8 GB) is not worth
            # that many locked pages for one copy: pageable memory, staged by the runtime
            h = torch.empty(t.shape, dtype=t.dtype, pin_memory=t.numel() * t.element_size() <= cap)
            h.copy_(t, non_blocking=True)
            t.record_stream(side)
            hosts.append(h)
        ev = torch.cuda.Event()
        ev.record(side)
    return hosts, ev


class R1_mAP_eval():
    def __init__(self, num_query, max_rank=50, feat_norm=True, reranking=False):
        super(R1_mAP_eval, self).__init__()
        self.num_query = num_query
        self.max_rank = max_rank
        self.feat_norm = feat_norm      # used as a truth value, like upstream ('yes' and 'no' both normalise)
        self.reranking = reranking
        self.distance_mode = _ops.GEMM_F32_EXACT   # _ops.GEMM_F16_SPLIT3 (<= 1e-6) / GEMM_F16_FAST (~1e-4): TEST.DISTANCE_MODE
        self.rerank_algo = _ops.RERANK_AUTO   # _ops.RERANK_SPARSE_SPLIT3: faster at large N, outputs within 1e-6
        self.last_rerank_stats = None

    def reset(self):
        self.feats = []
        self.pids = []
        self.camids = []

    def update(self, output):  # called once for each batch
        feat, pid, camid = output
        dev = _ops._lib.require_gpu()
        # own a device-resident fp32 copy: the caller may reuse its buffer
        self.feats.append(feat.detach().to(device=dev, dtype=torch.float32, copy=True))
        self.pids.extend(np.asarray(pid))
        self.camids.extend(np.asarray(camid))

    def compute(self):  # called after each epoch
        from mpreid import distributed as D
        if D.sharded_active():
            return self._compute_sharded()
        feats = torch.cat(self.feats, dim=0)
        _check_finite(feats)
        if self.feat_norm:
            print("The test feature is normalized")
            feats = _ops.l2_normalize(feats)
        # the features go to the host only because compute() returns them: their D2H copy (98 MB at Market-1501 scale) starts
        # NOW, on a side stream, and runs beside the distance GEMM / the re-ranking (round 6; it used to queue behind them)
        (h_feats,), feats_copied = _to_host_async([feats])
        qf = feats[:self.num_query]
        gf = feats[self.num_query:]
        q_pids = np.asarray(self.pids[:self.num_query])
        q_camids = np.asarray(self.camids[:self.num_query])
        g_pids = np.asarray(self.pids[self.num_query:])
        g_camids = np.asarray(self.camids[self.num_query:])
        if self.reranking:
            print('=> Enter reranking')
            dist, self.last_rerank_stats = re_ranking_device(qf, gf, k1=50, k2=15, lambda_value=0.3,
                                                             algo=getattr(self, "rerank_algo", 0))
            # the evaluator is called once per run: do not keep the re-ranking workspace (GBs at MSMT17 scale)
            # pinned beside the encoder's for the rest of the process
            _ops.release_workspaces("rerank")
        else:
            print('=> Computing DistMat with euclidean_distance')
            dist = _ops.euclidean_distance(qf, gf, mode=self.distance_mode)
        # ranking statistics on the GPU while the matrix is still resident; the matrix's own D2H copy runs on the side stream
        # BESIDE the ranking kernel.  (Measured, tools/evalrank_bench.py at Market-1501 shape: the kernel takes 0.15 ms alone
        # and 0.20 ms beside the copy; queueing the copy BEHIND it makes compute() 0.8 ms slower.  The 5.4 ms average of
        # round 5's rocprofv3 trace was the profiler serialising the two queues -- the kernel's interval there includes the
        # blit kernels it waited for.  MPREID_EVAL_D2H=behind orders the copy after the kernel: used for kernel traces only.)
        import os
        box = {}

        def start_copy():
            box["h"], box["ev"] = _to_host_async([dist])
        if os.environ.get("MPREID_EVAL_D2H") == "behind":
            cmc, mAP = eval_func_device(dist, q_pids, g_pids, q_camids, g_camids, after_launch=start_copy)
        else:
            start_copy()
            cmc, mAP = eval_func_device(dist, q_pids, g_pids, q_camids, g_camids)
        (h_dist,), copied = box["h"], box["ev"]
        feats_copied.synchronize()
        copied.synchronize()
        return cmc, mAP, h_dist.numpy(), self.pids, self.camids, h_feats[:self.num_query], h_feats[self.num_query:]

    def _compute_sharded(self):
        """compute() with one evaluator instance per rank of the default process group (one process per GPU; replaces the
        reference's nn.DataParallel branch, processor/processor.py:178-182; partition of SURVEY.md section 8e).

        Contract: `num_query` is the GLOBAL number of queries; rank r was update()d with ITS samples only, in global
        order -- queries shard_range(num_query, r, P) first, then gallery rows shard_range(num_gallery, r, P)
        (processor.do_inference shards the loader that way).  Steps: L2-normalise locally; ONE all-gather of the query
        features over xGMI; the rank's [nq, ng_local] column block of the distance matrix (its gallery shard) -- or, with
        re-ranking, the row-sharded phases of mpreid.distributed.re_ranking_sharded; ranking statistics on every rank's
        own query rows (eval_func_sharded); the blocks concatenated on the host of rank 0.  No floating-point reduction
        crosses ranks: rank 0 returns the 7-tuple of the single-process compute() byte for byte; the other ranks get the
        same cmc / mAP / pids / camids / qf, None for distmat and their own gallery shard for gf."""
        import torch.distributed  # noqa: F401  (ReduceOp)
        from mpreid import distributed as D
        rank, world = D.rank_world()
        dev = _ops._lib.require_gpu()
        nq = self.num_query
        q_lo, q_hi = D.shard_range(nq, rank, world)
        nql = q_hi - q_lo
        n_local = sum(f.shape[0] for f in self.feats)
        assert n_local >= nql, f"rank {rank}: {n_local} samples but {nql} of them must be its query shard"
        dim = torch.tensor([self.feats[0].shape[1] if self.feats else 0], dtype=torch.int64,
                           device="cpu" if D._pg().get_backend() == "gloo" else dev)
        D._pg().all_reduce(dim, op=torch.distributed.ReduceOp.MAX)
        feats = torch.cat(self.feats, dim=0) if self.feats else torch.empty((0, int(dim.item())), device=dev)
        _check_finite(feats, collective=True)
        if self.feat_norm:
            if rank == 0:
                print("The test feature is normalized")
            feats = _ops.l2_normalize(feats)
        meta = [None] * world     # labels (python ints): metadata, not the data path
        D._pg().all_gather_object(meta, ([int(p) for p in self.pids], [int(c) for c in self.camids], n_local - nql))
        ng_sizes = [m[2] for m in meta]
        ng = sum(ng_sizes)
        assert ng_sizes == D.shard_sizes(ng, world), (
            f"gallery shards {ng_sizes} are not shard_sizes({ng}, {world}): feed every rank its shard_range slice")
        q_sizes = D.shard_sizes(nq, world)
        pids = [p for m, k in zip(meta, q_sizes) for p in m[0][:k]] + [p for m, k in zip(meta, q_sizes) for p in m[0][k:]]
        camids = [c for m, k in zip(meta, q_sizes) for c in m[1][:k]] + [c for m, k in zip(meta, q_sizes) for c in m[1][k:]]
        q_pids, g_pids = np.asarray(pids[:nq]), np.asarray(pids[nq:])
        qf_local, gf_local = feats[:nql].contiguous(), feats[nql:].contiguous()
        qf = D.all_gather_rows(qf_local, nq)            # the RCCL all-gather of the query features
        if self.reranking:
            if rank == 0:
                print('=> Enter reranking')
            gf = D.all_gather_rows(gf_local, ng)        # every rank needs all rows of the N x N problem's operands
            rows = D.re_ranking_sharded(qf, gf, 50, 15, 0.3, algo=getattr(self, "rerank_algo", 0))   # [nql, ng]
            _ops.release_workspaces("rerank")
            cmc, mAP = eval_func_sharded(rows, q_pids[q_lo:q_hi], g_pids)
            distmat = D.gather_row_blocks_to_host(rows, dst=0)
            gf_out = gf
        else:
            if rank == 0:
                print('=> Computing DistMat with euclidean_distance')
            block = _ops.euclidean_distance(qf, gf_local, mode=self.distance_mode)     # [nq, ng_local]
            rows = D.column_to_row_blocks(block, nq, ng_sizes)                         # [nql, ng]
            cmc, mAP = eval_func_sharded(rows, q_pids[q_lo:q_hi], g_pids)
            distmat = D.gather_column_blocks_to_host(block, dst=0)                     # host concatenation on rank 0
            gf_host = D.gather_row_blocks_to_host(gf_local, dst=0)
            gf_out = torch.from_numpy(gf_host) if rank == 0 else gf_local
        return cmc, mAP, distmat, pids, camids, qf.cpu(), gf_out.cpu()
