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


def eval_func_device(dist, q_pids, g_pids, q_camids=None, g_camids=None, max_rank=50):
    """eval_func with the ranking done on the GPU (dist: device tensor [nq, ng] fp32, left on the device).

    Per query the kernel returns the positions of the relevant gallery items in the ascending (distance, index)
    order of the row — what the reference reads off np.argsort — in one pass over the row; CMC and AP are
    finished here in float64.  CMC is identical to eval_func's; AP sums the same terms (hits up to k) / k in a
    different order than numpy's pairwise reduction over the dense row, i.e. |delta mAP| ~ 1e-16."""
    import ctypes as C
    from mpreid import _lib
    dev = _lib.require_gpu()
    L = _lib.load()
    dist = dist.detach()
    assert dist.is_cuda and dist.dtype == torch.float32 and dist.dim() == 2 and dist.stride(1) == 1
    num_q, num_g = dist.shape
    q_pids = np.ascontiguousarray(q_pids, dtype=np.int64)
    g_pids = np.ascontiguousarray(g_pids, dtype=np.int64)
    if num_g < max_rank:
        max_rank = num_g
        print("Note: number of gallery samples is quite small, got {}".format(num_g))
    rcap = int(min(max(np.unique(g_pids, return_counts=True)[1].max(), 1), 2048))
    qp, gp = torch.from_numpy(q_pids).to(dev), torch.from_numpy(g_pids).to(dev)
    pos = torch.empty((num_q, rcap), dtype=torch.int32, device=dev)
    cnt = torch.empty(num_q, dtype=torch.int32, device=dev)
    _lib.check(L.mpreid_eval_rank_positions(C.c_void_p(dist.data_ptr()), dist.stride(0), num_q, num_g,
                                            C.c_void_p(qp.data_ptr()), C.c_void_p(gp.data_ptr()), rcap,
                                            C.c_void_p(pos.data_ptr()), C.c_void_p(cnt.data_ptr()), _lib.stream_ptr()),
               "mpreid_eval_rank_positions")
    pos, cnt = pos.cpu().numpy().astype(np.int64), cnt.cpu().numpy().astype(np.int64)
    over = np.nonzero(cnt < 0)[0]          # queries with more relevant items than the kernel handles: host ranking
    if over.size:
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
    num_valid_q = float(valid.sum())
    assert num_valid_q > 0, "Error: all query identities do not appear in gallery"
    pos, cnt = pos[valid], cnt[valid]
    first = pos[:, 0]
    cmc_rows = (np.arange(max_rank)[None, :] >= first[:, None]).astype(np.float32)
    all_cmc = cmc_rows.sum(0) / num_valid_q
    t = np.arange(1, pos.shape[1] + 1, dtype=np.float64)[None, :]
    terms = np.where(pos >= 0, t / np.maximum(pos + 1.0, 1.0), 0.0)
    all_AP = terms.sum(axis=1) / cnt
    return all_cmc, np.mean(all_AP)


class R1_mAP_eval():
    def __init__(self, num_query, max_rank=50, feat_norm=True, reranking=False):
        super(R1_mAP_eval, self).__init__()
        self.num_query = num_query
        self.max_rank = max_rank
        self.feat_norm = feat_norm      # used as a truth value, like upstream ('yes' and 'no' both normalise)
        self.reranking = reranking
        self.distance_mode = _ops.GEMM_F32_EXACT
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
        feats = torch.cat(self.feats, dim=0)
        if self.feat_norm:
            print("The test feature is normalized")
            feats = _ops.l2_normalize(feats)
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
        # ranking statistics on the GPU while the matrix is still resident; the matrix itself goes to the host
        # only because compute() returns it
        cmc, mAP = eval_func_device(dist, q_pids, g_pids, q_camids, g_camids)
        distmat = dist.cpu().numpy()
        return cmc, mAP, distmat, self.pids, self.camids, qf.cpu(), gf.cpu()
