"""Drop-in for the reference's ``utils/reranking.py`` (k-reciprocal re-ranking, Zhong et al. CVPR'17),
computed on the MI355X through libmpreid_hip.so.

Same call as the reference (utils/reranking.py:29):

    re_ranking(probFea, galFea, k1, k2, lambda_value, local_distmat=None, only_local=False)
        -> np.ndarray float32 [num_query, num_gallery]

probFea / galFea are torch tensors (any device); local_distmat an optional (nq+ng) x (nq+ng) array.
The algorithm and every fp16/fp32 rounding point follow the reference (see csrc/rerank.hip); the
dense N x N float16 ``V`` of the reference is replaced by sparse rows, which changes no result.
"""
import numpy as np
import torch

from mpreid import ops as _ops


def re_ranking_device(probFea, galFea, k1, k2, lambda_value, local_distmat=None, only_local=False, timing=False,
                      algo=_ops.RERANK_AUTO):
    """Same computation, result left on the GPU; returns (tensor [nq, ng], stats dict).  algo: mpreid.ops.RERANK_AUTO
    (bit-parity, the default) ... RERANK_SPARSE_SPLIT3 (blend-term distances from the fp16 matrix cores, outputs within
    1e-6, ranks identical; the large-N option)."""
    return _ops.re_ranking(probFea, galFea, k1, k2, lambda_value, local_distmat=local_distmat,
                           only_local=only_local, timing=timing, algo=algo)


def re_ranking(probFea, galFea, k1, k2, lambda_value, local_distmat=None, only_local=False):
    if isinstance(probFea, np.ndarray):
        probFea = torch.from_numpy(probFea)
    if isinstance(galFea, np.ndarray):
        galFea = torch.from_numpy(galFea)
    final_dist, _ = re_ranking_device(probFea, galFea, k1, k2, lambda_value, local_distmat, only_local)
    return final_dist.cpu().numpy()
