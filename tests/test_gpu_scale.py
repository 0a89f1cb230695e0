"""BASELINE.json configs[3] (synthetic 20 000 query x 80 000 gallery, 768-d, gallery sharded over up to 8 GPUs) and
configs[4] (MSMT17 shape: N = 93 820, D = 1280) at their FULL sizes on one MI355X.

The oracle cannot run there (dense N x N on the host), so the checks are the size-independent ones:
  * the distance matrix written as column blocks by 1 / 2 / 8 virtual shards equals the single call bit for bit, and
    sampled entries equal an fp64 evaluation within 1e-5 (north_star's entry bound);
  * re-ranking: lambda = 1 collapses to the normalised distance block exactly; idempotence; the row-sharded phases
    (8 virtual ranks) give the same bits as the single call; mAP improves over the plain ranking;
  * and the same code path is compared bit for bit with the ORACLE on a 1/8 sub-sample of the very same features."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from mpreid import ops as o
    yield o
    o.release_workspaces()
    torch.cuda.empty_cache()


def _features(n, d, sigma, seed):
    """SURVEY.md §8d clustered features, generated on the device (300-480 MB), normalised by the HIP kernel"""
    from mpreid import ops as o
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    n_ids = n // 20
    cent = torch.randn((n_ids, d), generator=g, device="cuda")
    pid = torch.randint(0, n_ids, (n,), generator=g, device="cuda")
    x = cent[pid] + sigma * torch.randn((n, d), generator=g, device="cuda")
    return o.l2_normalize(x), pid.cpu().numpy()


def test_synth_20k_x_80k_distmat_sharded_equals_single(ops):
    from mpreid import distributed as D
    nq, ng, d = 20000, 80000, 768
    f, _ = _features(nq + ng, d, 3.0, 1234)
    q, g = f[:nq], f[nq:]
    single = ops.euclidean_distance(q, g)                      # 6.4 GB
    for world in (2, 8):
        sharded = torch.empty_like(single)
        for r in range(world):                                  # every virtual rank writes its column block
            lo, hi = D.shard_range(ng, r, world)
            ops.euclidean_distance(q, g[lo:hi], out=sharded, col_offset=lo)
        assert torch.equal(single, sharded), world
        # and as separate per-rank blocks concatenated on the host side (north_star), on a row sample
        rows = torch.arange(0, nq, 997, device="cuda")
        blocks = [ops.euclidean_distance(q[rows], g[slice(*D.shard_range(ng, r, world))]) for r in range(world)]
        assert torch.equal(torch.cat(blocks, dim=1), single[rows])
        del sharded
    # entries against fp64 (|delta| <= 1e-5: north_star)
    rng = np.random.default_rng(0)
    qi, gi = rng.integers(0, nq, 4096), rng.integers(0, ng, 4096)
    qd, gd = q[qi].double(), g[gi].double()
    want = (qd * qd).sum(1) + (gd * gd).sum(1) - 2.0 * (qd * gd).sum(1)
    got = single[torch.from_numpy(qi).cuda(), torch.from_numpy(gi).cuda()].double()
    assert float((got - want).abs().max()) <= 1e-5
    # the fp16 one-pass speed mode is NOT a parity mode: its error is reported, bounded at the documented level
    fast = ops.euclidean_distance(q[:2048], g[:8192], mode=ops.GEMM_F16_FAST)
    err = float((fast - single[:2048, :8192]).abs().max())
    assert 1e-6 < err < 1e-3, err


@pytest.mark.parametrize("name,nq,ng,d,sigma", [("synth_100k", 20000, 80000, 768, 3.0),
                                                ("msmt17", 11659, 82161, 1280, 3.5)])
def test_rerank_full_size_properties(ops, name, nq, ng, d, sigma):
    from mpreid import distributed as D
    from utils.metrics import eval_func
    N = nq + ng
    f, pid = _features(N, d, sigma, 4321)
    q, g = f[:nq], f[nq:]
    # lambda = 1: final_dist == (original_dist / colmax).T[:nq, nq:] exactly
    r1, _ = ops.re_ranking(q, g, 50, 15, 1.0)
    colmax = torch.empty(nq, device="cuda")
    for s in range(0, nq, 4096):                                 # max over column i of D == max over row i (symmetric)
        e = min(nq, s + 4096)
        colmax[s:e] = ops.euclidean_distance(f[s:e], f).max(dim=1).values
    for s in range(0, nq, 4096):
        e = min(nq, s + 4096)
        want = ops.euclidean_distance(f[s:e], g) / colmax[s:e, None] * np.float32(1.0)
        assert torch.equal(r1[s:e], want), s
    del r1, want
    ra, st = ops.re_ranking(q, g, 50, 15, 0.3, timing=True)
    assert st["algo"] == ops.RERANK_SPARSE and st["fallback_rows"] < N // 16
    rb, _ = ops.re_ranking(q, g, 50, 15, 0.3)
    assert torch.equal(ra, rb)                                   # idempotent / run-to-run deterministic
    rb, std = ops.re_ranking(q, g, 50, 15, 0.3, timing=True, algo=ops.RERANK_DENSE)   # the N x N algorithm (35-40 GB)
    assert torch.equal(ra, rb), "sparse and dense algorithms differ"
    print(name, "dense stats", std)
    del rb
    assert st["vqe_nnz"] > st["v_nnz"] > N * 10 and st["jaccard_pairs"] > 0
    print(name, "re-rank stats", st)
    ops.release_workspaces()
    torch.cuda.empty_cache()
    rv = D.re_ranking_virtual(q, g, 50, 15, 0.3, 8)             # the 8-GPU phases, one virtual rank after the other
    assert torch.equal(ra, rv)
    del rv
    d_plain = ops.euclidean_distance(q[:2000], g).cpu().numpy()
    _, map_plain = eval_func(d_plain, pid[:2000], pid[nq:], None, None)
    _, map_rr = eval_func(ra[:2000].cpu().numpy(), pid[:2000], pid[nq:], None, None)
    assert map_rr > map_plain, (map_plain, map_rr)
    del ra
    # 1/8 sub-sample of the same features: bit for bit against the oracle
    fs = torch.cat([q[::8], g[::8]]).contiguous()
    nqs = q[::8].shape[0]
    got, _ = ops.re_ranking(fs[:nqs], fs[nqs:], 50, 15, 0.3)
    fh = fs.cpu().numpy()
    want = orc.re_ranking(fh[:nqs], fh[nqs:], 50, 15, 0.3)
    assert np.array_equal(got.cpu().numpy(), want)
