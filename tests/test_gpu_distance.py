"""GPU parity: L2-normalise, squared norms, euclidean / cosine distance matrices (C ABI via ctypes).

Bar: the exact fp32 mode is BIT-EXACT against the oracle (same k-ascending fmaf chain) and within
1e-5 of the reference's golden output; the one-pass fp16 mode is within 3e-4 (stated, not parity)."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from mpreid import ops as o
    return o


def _feat(n, d, seed, normalize=True):
    from mpreid import synth
    return synth.clustered_features(n, d, 3.0, seed=seed, per_id=10, normalize=normalize)[0]


@pytest.mark.parametrize("n,d", [(1, 1), (5, 3), (64, 64), (130, 100), (257, 768), (300, 1280), (1000, 2048)])
def test_sqnorm_and_normalize_bit_exact(ops, n, d):
    x = _feat(n, d, seed=n + d, normalize=False)
    got = ops.sqnorm(torch.from_numpy(x)).cpu().numpy()
    assert np.array_equal(got, orc.sqnorm(x))
    got = ops.l2_normalize(torch.from_numpy(x)).cpu().numpy()
    assert np.array_equal(got, orc.l2_normalize(x))


def test_normalize_zero_row(ops):
    x = np.zeros((3, 16), np.float32)
    x[1] = 1e-20
    got = ops.l2_normalize(torch.from_numpy(x)).cpu().numpy()
    assert np.array_equal(got, orc.l2_normalize(x)) and np.isfinite(got).all()


@pytest.mark.parametrize("nq,ng,d", [(1, 1, 4), (3, 200, 7), (129, 257, 100), (128, 128, 16), (64, 256, 1280),
                                      (300, 515, 768), (500, 1000, 1280)])
def test_euclid_exact_bit_exact_vs_oracle(ops, nq, ng, d):
    f = _feat(nq + ng, d, seed=nq * 7 + ng)
    got = ops.euclidean_distance(torch.from_numpy(f[:nq]), torch.from_numpy(f[nq:])).cpu().numpy()
    want = orc.euclidean_distance(f[:nq], f[nq:])
    assert got.shape == want.shape and np.array_equal(got, want)


def test_euclid_unnormalised_bit_exact(ops):
    f = _feat(400, 1280, seed=5, normalize=False)
    got = ops.euclidean_distance(torch.from_numpy(f[:100]), torch.from_numpy(f[100:])).cpu().numpy()
    assert np.array_equal(got, orc.euclidean_distance(f[:100], f[100:]))


def test_euclid_symmetric_bits(ops):
    f = torch.from_numpy(_feat(700, 768, seed=9))
    d = ops.euclidean_distance(f, f)
    assert torch.equal(d, d.t().contiguous())


def test_distance_vs_reference_golden(ops, golden):
    from mpreid import synth
    g = golden("distance.npz")
    feat, _ = synth.clustered_features(int(g["n"]), int(g["dim"]), float(g["sigma"]), seed=int(g["seed"]),
                                       per_id=int(g["per_id"]))
    nq = int(g["nq"])
    q, ga = torch.from_numpy(feat[:nq]), torch.from_numpy(feat[nq:])
    d = ops.euclidean_distance(q, ga).cpu().numpy()
    assert np.abs(d - g["euclid"]).max() < 1e-5
    c = ops.cosine_similarity(q, ga).cpu().numpy()
    assert np.abs(c - g["cosine"]).max() < 1e-5
    assert np.abs(c - orc.cosine_similarity(feat[:nq], feat[nq:])).max() < 2e-6
    # drop-in module functions return numpy float32
    from utils import metrics
    d2 = metrics.euclidean_distance(q, ga)
    assert isinstance(d2, np.ndarray) and d2.dtype == np.float32 and np.array_equal(d2, d)
    assert np.abs(metrics.cosine_similarity(q, ga) - g["cosine"]).max() < 1e-5


def test_euclid_fp16_fast_tolerance(ops):
    f = _feat(900, 768, seed=3)
    q, g = torch.from_numpy(f[:300]), torch.from_numpy(f[300:])
    fast = ops.euclidean_distance(q, g, mode=ops.GEMM_F16_FAST).cpu().numpy()
    want = orc.euclidean_distance(f[:300], f[300:])
    err = np.abs(fast - want).max()
    assert err < 3e-4, err


def test_column_block_output(ops):
    """a rank writes its gallery shard into a column block of a wider matrix (ldo > ng)"""
    f = _feat(500, 256, seed=4)
    q = torch.from_numpy(f[:100])
    full = torch.zeros((100, 400), dtype=torch.float32, device="cuda")
    ops.euclidean_distance(q, torch.from_numpy(f[100:300]), out=full, col_offset=0)
    ops.euclidean_distance(q, torch.from_numpy(f[300:]), out=full, col_offset=200)
    assert np.array_equal(full.cpu().numpy(), orc.euclidean_distance(f[:100], f[100:]))


def test_gemm_f16_exact_integers(ops):
    """asymmetric small-integer operands: fp16 products and fp32 sums are exact, so any layout
    mistake (row/col swap, wrong k order inside a fragment, swizzle) shows up as a mismatch."""
    rng = np.random.default_rng(0)
    m, n, k = 256, 384, 192
    a = rng.integers(-4, 5, size=(m, k)).astype(np.float32)
    b = rng.integers(-4, 5, size=(n, k)).astype(np.float32)
    b[:, 0] += np.arange(n) % 3  # break symmetry
    c = ops.gemm_f16_nt(torch.from_numpy(a).half().cuda(), torch.from_numpy(b).half().cuda()).cpu().numpy()
    assert np.array_equal(c, a @ b.T)


@pytest.mark.parametrize("m,n,k", [(256, 256, 64), (512, 768, 192), (1024, 512, 768), (256, 1024, 3072)])
def test_gemm_f16_big_kernel_exact_integers(m, n, k):
    """the 256x256 ring-pipelined kernel (forced with MPREID_TUNE=gemm_big=2 in a child process)"""
    import os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent(f"""
        import sys, numpy as np, torch
        sys.path[:0] = [{root!r}, {root!r} + "/mp-reid_amd"]
        from mpreid import ops
        rng = np.random.default_rng(1)
        a = rng.integers(-4, 5, size=({m}, {k})).astype(np.float32)
        b = rng.integers(-4, 5, size=({n}, {k})).astype(np.float32)
        b[:, 0] += np.arange({n}) % 3
        for rep in range(3):
            c = ops.gemm_f16_nt(torch.from_numpy(a).half().cuda(), torch.from_numpy(b).half().cuda()).cpu().numpy()
            assert np.array_equal(c, a @ b.T), np.abs(c - a @ b.T).max()
        print("ok")
    """)
    env = dict(os.environ, MPREID_TUNE="gemm_big=2")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def test_full_size_properties(ops):
    """Market-1501 shape (3368 x 15913 x 1280): size-independent checks instead of the oracle."""
    from mpreid import synth
    f, _ = synth.clustered_features(19281, 1280, 3.5, seed=1234)
    ft = torch.from_numpy(f).cuda()
    q, g = ft[:3368], ft[3368:]
    d = ops.euclidean_distance(q, g)
    assert d.shape == (3368, 15913) and torch.isfinite(d).all()
    # a row block recomputed alone is bit-identical (tile position independence)
    d2 = ops.euclidean_distance(q[1000:1100], g[5000:6000])
    assert torch.equal(d2, d[1000:1100, 5000:6000])
    # against fp64 on a random sample of entries
    idx_q = torch.randint(0, 3368, (2000,), device="cuda")
    idx_g = torch.randint(0, 15913, (2000,), device="cuda")
    ref = ((q[idx_q].double() - g[idx_g].double()) ** 2).sum(1)
    assert (d[idx_q, idx_g].double() - ref).abs().max().item() < 2e-6


@pytest.mark.parametrize("epi", [1, 2, 3])
def test_gemm_kernels_agree_across_variants(epi):
    """the three fp16 GEMM kernels (128x128, 256x256 ring, persistent 256x128) must give identical
    results for the fused epilogues (same k order inside every MFMA chain => bit identical)"""
    import os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent(f"""
        import sys, ctypes as C, numpy as np, torch
        sys.path[:0] = [{root!r}, {root!r} + "/mp-reid_amd"]
        from mpreid import _lib
        L = _lib.load(); dev = _lib.require_gpu()
        torch.manual_seed(0)
        m, n, k, epi = 8192, 2048, 768, {epi}
        A = (torch.rand((m, k), device=dev) - 0.5).half(); W = ((torch.rand((n, k), device=dev) - 0.5) * 0.1).half()
        bias = torch.randn(n, device=dev)
        base = torch.randn((m, n), device=dev)
        out = base.clone() if epi == 2 else torch.zeros((m, n), device=dev, dtype=torch.float16)
        _lib.check(L.mpreid_gemm_f16_nt_ex(C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(out.data_ptr()),
                   C.c_void_p(bias.data_ptr()), m, n, k, epi, _lib.stream_ptr()), "gemm")
        torch.cuda.synchronize()
        np.save(sys.argv[1], out.float().cpu().numpy())
    """)
    outs = []
    for mode in ("0", "2", "3"):
        path = f"/tmp/gemm_variant_{epi}_{mode}.npy"
        r = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, MPREID_TUNE="gemm_big=" + mode),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(np.load(path))
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    # and against an fp32 torch reference of the same op
    torch.manual_seed(0)
    m, n, k = 8192, 2048, 768
    A = (torch.rand((m, k), device="cuda") - 0.5).half(); W = ((torch.rand((n, k), device="cuda") - 0.5) * 0.1).half()
    bias = torch.randn(n, device="cuda"); base = torch.randn((m, n), device="cuda")
    ref = A.float() @ W.float().t() + bias
    if epi == 2:
        ref = base + ref
    if epi == 3:
        ref = ref * torch.sigmoid(1.702 * ref)
    err = (torch.from_numpy(outs[0]).cuda() - ref).abs().max().item()
    assert err < (2e-3 if epi == 2 else 2e-2), err


# ---------------------------------------------------------------------------------------------
# 3-term fp16 split mode (MPREID_GEMM_F16_SPLIT3): parity-grade on the fp16 matrix cores
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nq,ng,d", [(300, 515, 768), (500, 1000, 1280), (129, 257, 100), (1024, 2048, 768), (64, 256, 2048)])
def test_euclid_split3_within_1e6_of_oracle(ops, nq, ng, d):
    f = _feat(nq + ng, d, seed=nq * 7 + ng)
    got = ops.euclidean_distance(torch.from_numpy(f[:nq]), torch.from_numpy(f[nq:]), mode=ops.GEMM_F16_SPLIT3).cpu().numpy()
    want = orc.euclidean_distance(f[:nq], f[nq:])
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 1e-6, np.abs(got - want).max()
    # against an fp64 evaluation it is as accurate as the exact fp32 chain itself
    q64, g64 = f[:nq].astype(np.float64), f[nq:].astype(np.float64)
    ref = (q64 * q64).sum(1)[:, None] + (g64 * g64).sum(1)[None, :] - 2.0 * q64 @ g64.T
    assert np.abs(got - ref).max() <= 1.5 * max(np.abs(want - ref).max(), 4e-7)


def test_split3_unnormalised_and_cosine(ops, golden):
    from mpreid import synth
    f = _feat(400, 1280, seed=5, normalize=False) * np.float32(37.0)       # row norms ~ 4e3: per-row scaling at work
    got = ops.euclidean_distance(torch.from_numpy(f[:100]), torch.from_numpy(f[100:]), mode=ops.GEMM_F16_SPLIT3).cpu().numpy()
    want = orc.euclidean_distance(f[:100], f[100:])
    assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max()
    g = golden("distance.npz")
    feat, _ = synth.clustered_features(int(g["n"]), int(g["dim"]), float(g["sigma"]), seed=int(g["seed"]),
                                       per_id=int(g["per_id"]))
    nq = int(g["nq"])
    q, ga = torch.from_numpy(feat[:nq]), torch.from_numpy(feat[nq:])
    assert np.abs(ops.euclidean_distance(q, ga, mode=ops.GEMM_F16_SPLIT3).cpu().numpy() - g["euclid"]).max() < 1e-5
    assert np.abs(ops.cosine_similarity(q, ga, mode=ops.GEMM_F16_SPLIT3).cpu().numpy() - g["cosine"]).max() < 1e-5


@pytest.mark.parametrize("case", ["50_15_0.3", "20_6_0.3", "7_3_0.5"])
def test_split3_gives_the_oracles_neighbour_tables_on_the_goldens(ops, golden, case):
    """initial_rank[:, :k1+1] computed from split-3 distances == the oracle's (the goldens are tie-free in the
    top-(k1+2), tests/golden/make_goldens.py)"""
    g = golden("rerank.npz")
    k1 = int(case.split("_")[0])
    k2, lam = int(case.split("_")[1]), float(case.split("_")[2])
    feat = g["feat"]
    nq = int(g["nq"])
    _, orank, _, _ = orc.re_ranking(feat[:nq], feat[nq:], k1, k2, lam, debug=True)
    ft = torch.from_numpy(feat).cuda()
    d = ops.euclidean_distance(ft, ft, mode=ops.GEMM_F16_SPLIT3)
    o = (d / d.max(dim=0).values[None, :]).t().contiguous()            # utils/reranking.py:46
    rank = torch.argsort(o, dim=1, stable=True)[:, :k1 + 1].cpu().numpy()
    assert np.array_equal(rank, orank)


def test_stored_distance_kernels_bit_identical():
    """the persistent 256 x 256 kernel's stored-distance path (paired DMA, early prologue with store-tolerant waits,
    interior-tile fast epilogue + the general one on ragged edges) against the 128 x 128 kernel (MPREID_TUNE=gemm_big=0, latched
    per process: child processes): same k order inside every accumulator => the outputs must agree bit for bit.  Shapes:
    Market-1501 (ragged rows and columns), K = 128 and K = 64 (pipelines shorter than the early-prologue threshold), the
    3-term split (row / column scales, K = 3 d), a single tile row, and a 20k-class square"""
    import os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent(f"""
        import sys, hashlib, numpy as np, torch
        sys.path[:0] = [{root!r}, {root!r} + "/mp-reid_amd"]
        from mpreid import ops, synth
        res = []
        for nq, ng, d, mode in ((3368, 15913, 768, "fast"), (3368, 15913, 768, "split3"), (2900, 11111, 100, "fast"),
                                (4096, 8192, 64, "fast"), (3000, 11000, 1280, "split3"), (200, 9000, 768, "fast"),
                                (12000, 12000, 768, "fast")):
            q = torch.from_numpy(synth.clustered_features(nq, d, 3.0, seed=5, per_id=10)[0]).cuda()
            g = torch.from_numpy(synth.clustered_features(ng, d, 3.0, seed=6, per_id=10)[0]).cuda()
            m = ops.GEMM_F16_FAST if mode == "fast" else ops.GEMM_F16_SPLIT3
            hs = set()
            for _ in range(3):
                o = ops.euclidean_distance(q, g, mode=m)
                torch.cuda.synchronize()
                hs.add(hashlib.sha256(o.cpu().numpy().tobytes()).hexdigest())
            assert len(hs) == 1, (nq, ng, d, mode, "not reproducible")
            res.append(hs.pop())
        print("HASHES " + " ".join(res))
    """)
    outs = []
    # ... and (round 5) the two-workgroups-per-CU 256 x 128 kernel in its two-tensor form (dist_p2_full=2: whenever the padded
    # sizes allow -- every shape here: rows are padded to 256, columns to 256)
    for tune in ("gemm_big=0", "gemm_big=2", "gemm_big=2,dist_p2_full=2"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MPREID_TUNE=tune), capture_output=True,
                           text=True, timeout=900)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append([ln for ln in r.stdout.splitlines() if ln.startswith("HASHES ")][0].split()[1:])
    assert outs[0] == outs[1], list(zip(outs[0], outs[1]))
    assert outs[0] == outs[2], list(zip(outs[0], outs[2]))


# ---- all-pairs distances of ONE set: euclidean_distance(f, f) computes the upper-triangular tiles and mirrors them ----
@pytest.mark.parametrize("n,d,big,p2", [(20000, 768, "1", "1"), (20000, 768, "1", "0"), (16384, 128, "1", "1"), (16384, 128, "1", "0"),
                                         (1000, 192, "2", "2"), (1000, 192, "2", "0"), (1000, 192, "1", "1"), (4133, 1280, "2", "2"),
                                         (4133, 1280, "2", "0"), (300, 96, "2", "2"), (6000, 2048, "1", "1"), (6000, 768, "1", "3")])
def test_all_pairs_symmetric_path_same_bits_as_full_computation(n, d, big, p2):
    """euclidean_distance(f, f, F16_FAST) -- same pointer, so the persistent kernel takes its symmetric form (tiles on or
    above the diagonal, every off-diagonal tile stored twice) -- equals euclidean_distance(f, copy of f) -- the full
    computation -- bit for bit: 79 x 79 tiles with ragged edges (blocked walk), 64 x 64 tiles (XCD-owned walk), a grid
    smaller than the chip (strided walk; also the 128 x 128 kernel, which ignores the flag), and a matrix embedded in a
    wider one (ldo > n).  The 3-term split mode: within 1e-6 of the exact chain, off-diagonal tiles exactly mirrored."""
    import os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent(f"""
        import sys, numpy as np, torch
        sys.path[:0] = [{root!r}, {root!r} + "/mp-reid_amd"]
        from mpreid import ops, synth
        f, _ = synth.clustered_features({n}, {d}, 3.0, seed=11)
        ft = torch.from_numpy(f).cuda()
        ft2 = ft.clone()
        full = ops.euclidean_distance(ft, ft2, mode=ops.GEMM_F16_FAST)
        for rep in range(12):   # (also a race screen: the two workgroups of a CU drift through every relative phase)
            sym = ops.euclidean_distance(ft, ft, mode=ops.GEMM_F16_FAST)
            assert torch.equal(sym, full), float((sym - full).abs().max())
        assert torch.equal(sym, sym.t())
        wide = torch.full(({n}, {n} + 40), -7.0, device="cuda")
        ops.euclidean_distance(ft, ft, mode=ops.GEMM_F16_FAST, out=wide, col_offset=24)
        assert torch.equal(wide[:, 24:24 + {n}], full) and bool((wide[:, :24] == -7).all()) and bool((wide[:, 24 + {n}:] == -7).all())
        s3 = ops.euclidean_distance(ft, ft, mode=ops.GEMM_F16_SPLIT3)
        assert float((s3 - s3.t()).abs().max()) <= 5e-7     # exactly mirrored off the diagonal tiles; inside them the 3-term sum is computed both ways
        if {n} >= 4096 or "{big}" == "2" or "{p2}" == "2":   # a symmetric kernel ran (the 128 x 128 kernel computes every tile)
            assert torch.equal(s3[512:768, 0:256], s3[0:256, 512:768].t())
        m = min({n}, 3000)
        ex = ops.euclidean_distance(ft[:m].contiguous(), ft)
        s3f = ops.euclidean_distance(ft, ft2, mode=ops.GEMM_F16_SPLIT3)
        e_sym, e_full = float((s3[:m] - ex).abs().max()), float((s3f[:m] - ex).abs().max())
        assert e_sym <= max(1e-6, 1.1 * e_full), (e_sym, e_full)     # (the tail of 6e7 entries can pass 1e-6 by a hair, either way)
        assert float((s3 - s3f).abs().max()) <= 5e-7
        print("ok")
    """)
    # dist_sym_p2: 0 = the one-workgroup-per-CU kernel's symmetric instance, 1 = the two-workgroups-per-CU kernel from 16
    # tile rows on (the default), 2 = always, 3 = also for the 3-term split operands (which stay on the first by default)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MPREID_TUNE="gemm_big=" + big + ",dist_sym_p2=" + p2),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
