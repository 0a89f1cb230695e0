"""The sharded evaluator and the sharded re-ranking END TO END through the "nccl" branches of mpreid/distributed.py with
P = 2, 3 and 8 VIRTUAL ranks on one GPU (tests/emulated_group.py: threads of this process, collectives delivered by device
copies, RCCL's requirements asserted).  The gloo-staged multi-process tests (tests/test_gpu_distributed.py) take the gloo
branches; the one-rank RCCL test runs the nccl branches with a single split.  Here the ragged split / padding arithmetic of
the nccl branches runs on device tensors with P > 1.  Bar: rank 0's 7-tuple == the single-process compute() byte for byte.
Reference: processor/processor.py:178-182 (its multi-device branch), utils/metrics.py:110-134, utils/reranking.py:29-100."""
import numpy as np
import pytest
import torch

from emulated_group import EmulatedWorld

pytestmark = pytest.mark.gpu

CASES = [(900, 150, 192, False), (2600, 500, 128, True), (643, 41, 64, True), (700, 5, 64, False), (4100, 800, 96, True)]


def _single(n, nq, d, rerank):
    from mpreid import synth
    from utils.metrics import R1_mAP_eval
    f, pid = synth.clustered_features(n, d, 2.5, seed=77 + n, per_id=6, normalize=False)
    cam = synth.labels_for(n)
    ev = R1_mAP_eval(nq, max_rank=50, feat_norm='yes', reranking=rerank)
    ev.reset()
    ev.update((torch.from_numpy(f).cuda(), tuple(int(p) for p in pid), tuple(int(c) for c in cam)))
    return f, pid, cam, ev.compute()


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("case", CASES)
def test_evaluator_through_the_rccl_branches_with_virtual_ranks(world, case):
    from mpreid import distributed as D
    from utils.metrics import R1_mAP_eval
    n, nq, d, rerank = case
    f, pid, cam, want = _single(n, nq, d, rerank)
    W = EmulatedWorld(world)

    def rank_fn(r):
        q_lo, q_hi = D.shard_range(nq, r, world)
        g_lo, g_hi = D.shard_range(n - nq, r, world)
        idx = list(range(q_lo, q_hi)) + list(range(nq + g_lo, nq + g_hi))      # this rank's samples, global order
        ev = R1_mAP_eval(nq, max_rank=50, feat_norm='yes', reranking=rerank)
        ev.reset()
        for s in range(0, len(idx), 64):
            sel = idx[s:s + 64]
            ev.update((torch.from_numpy(f[sel]).cuda(), tuple(int(p) for p in pid[sel]), tuple(int(c) for c in cam[sel])))
        return ev.compute()

    res = W.run(rank_fn)
    cmc, mAP, distmat, pids, camids, qf, gf = res[0]
    assert np.array_equal(cmc, want[0]) and cmc.dtype == want[0].dtype and float(mAP) == float(want[1])
    assert distmat.dtype == np.float32 and np.array_equal(distmat, want[2])
    assert list(pids) == [int(p) for p in want[3]] and list(camids) == [int(c) for c in want[4]]
    assert np.array_equal(qf.numpy(), want[5].numpy()) and np.array_equal(gf.numpy(), want[6].numpy())
    for r in range(1, world):
        assert res[r][2] is None and np.array_equal(res[r][0], want[0]) and float(res[r][1]) == float(want[1])
    # the collectives an 8-GPU run issues were all exercised with P > 1
    assert "all_gather_into_tensor" in W.log and "gather" in W.log and "all_gather_object" in W.log
    if not rerank:
        assert "all_to_all_single" in W.log


@pytest.mark.parametrize("world,n,nq,d,k1,k2", [(8, 6000, 1200, 256, 50, 15), (3, 4100, 800, 128, 20, 6), (5, 3000, 2999, 64, 30, 40),
                                                 (4, 900, 7, 64, 20, 6), (2, 700, 100, 128, 10, 1)])
def test_sharded_rerank_through_the_rccl_branches_with_virtual_ranks(world, n, nq, d, k1, k2, monkeypatch):
    """mpreid.distributed.re_ranking_sharded itself (not the virtual-rank loop of re_ranking_virtual): rank table, CSR
    all-gathers of V / V_qe, the column-sharded index exchange of phase 4, row blocks to the host -- P ranks, ragged shards.
    Both branches of phase 4 of the REAL function: the column-sharded index build (default) and MPREID_RR_FULL_INDEX=1, the
    documented switch back to round 4's form (every rank builds the whole index, no index exchange; advisor r5: the switch
    was only read by re_ranking_virtual) -- same bits, and the switch really removes the three index collectives."""
    from mpreid import distributed as D, ops, synth
    f, _ = synth.clustered_features(n, d, 2.5, seed=n + world, per_id=20)
    q, g = torch.from_numpy(f[:nq]).cuda(), torch.from_numpy(f[nq:]).cuda()
    single, _ = ops.re_ranking(q, g, k1, k2, 0.3)
    calls = {}
    for full_index in (False, True):
        if full_index:
            monkeypatch.setenv("MPREID_RR_FULL_INDEX", "1")
        else:
            monkeypatch.delenv("MPREID_RR_FULL_INDEX", raising=False)
        W = EmulatedWorld(world)

        def rank_fn(r):
            rows = D.re_ranking_sharded(q, g, k1, k2, 0.3)
            q_lo, q_hi = D.shard_range(nq, r, world)
            assert rows.shape == (q_hi - q_lo, n - nq)
            assert torch.equal(rows, single[q_lo:q_hi]), (r, full_index)
            return D.gather_row_blocks_to_host(rows, dst=0)

        res = W.run(rank_fn)
        assert np.array_equal(res[0], single.cpu().numpy()), full_index
        assert W.log.count("all_gather_into_tensor") >= (3 if full_index else 5)
        calls[full_index] = len(W.log)
    assert calls[False] > calls[True], calls   # counts, packed pieces, boundary rows: exchanged only by the sharded build
