#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by importing the REFERENCE in place.

Run in the build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_goldens.py

What is imported from the reference (nothing is copied into this repo):
  * /root/reference/utils/metrics.py    euclidean_distance, cosine_similarity, eval_func, R1_mAP_eval
  * /root/reference/utils/reranking.py  re_ranking
  * /root/reference/model/clip/model.py VisionTransformer (loaded by file path so that
    model/__init__.py, which needs torchvision/timm, is never executed)

The fixtures hold INPUTS (seeded, from mp-reid_amd/mpreid/synth.py) and the reference's OUTPUTS.
ViT weights are not stored: they are regenerated from the seed by synth.vit_state_dict().
"""
import importlib.util
import io
import os
import sys
import contextlib
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(ROOT, "mp-reid_amd"))
from mpreid import synth  # noqa: E402  (our own seeded generators)

# --- the reference, imported in place -------------------------------------------------------
sys.path.insert(0, REF)
# our package dir also has a 'utils' package: make sure 'utils' resolves to the reference here
sys.path.remove(os.path.join(ROOT, "mp-reid_amd"))
for m in [k for k in sys.modules if k == "utils" or k.startswith("utils.")]:
    del sys.modules[m]
from utils import metrics as ref_metrics      # noqa: E402
from utils import reranking as ref_reranking  # noqa: E402
assert ref_metrics.__file__.startswith(REF), ref_metrics.__file__

spec = importlib.util.spec_from_file_location("ref_clip_model", os.path.join(REF, "model/clip/model.py"))
ref_clip = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_clip)

warnings.filterwarnings("ignore")
torch.manual_seed(0)
torch.set_num_threads(8)


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path)/1024:.0f} KiB  " +
          " ".join(f"{k}{tuple(np.shape(v))}" for k, v in arrs.items()))


# --------------------------------------------------------------------------------------------
# (1) euclidean_distance / cosine_similarity, (2) eval_func
# --------------------------------------------------------------------------------------------
def gen_distance():
    feat, pid = synth.clustered_features(320, 1280, sigma=3.5, seed=11, per_id=8)
    nq = 64
    q, g = torch.from_numpy(feat[:nq]), torch.from_numpy(feat[nq:])
    d_e = ref_metrics.euclidean_distance(q, g)
    d_c = ref_metrics.cosine_similarity(q, g)
    # un-normalised inputs as well (feat_norm off path)
    raw, _ = synth.clustered_features(320, 1280, sigma=3.5, seed=11, per_id=8, normalize=False)
    d_e_raw = ref_metrics.euclidean_distance(torch.from_numpy(raw[:nq]), torch.from_numpy(raw[nq:]))
    d_c_raw = ref_metrics.cosine_similarity(torch.from_numpy(raw[:nq]), torch.from_numpy(raw[nq:]))
    save("distance.npz", seed=11, nq=nq, sigma=3.5, per_id=8, n=320, dim=1280,
         euclid=d_e, cosine=d_c, euclid_raw=d_e_raw, cosine_raw=d_c_raw)

    # eval_func: make query 5 an identity that is absent from the gallery
    q_pid = pid[:nq].copy()
    g_pid = pid[nq:].copy()
    q_pid[5] = 10_000
    cam = synth.labels_for(320)
    cmc, mAP = quiet(ref_metrics.eval_func, d_e, q_pid, g_pid, cam[:nq], cam[nq:])
    # small-gallery branch (num_g < max_rank)
    cmc_s, mAP_s = quiet(ref_metrics.eval_func, d_e[:, :30], q_pid, g_pid[:30], cam[:nq], cam[nq:nq + 30])
    save("eval_func.npz", q_pid=q_pid, g_pid=g_pid, q_cam=cam[:nq], g_cam=cam[nq:],
         cmc=cmc, mAP=np.float64(mAP), cmc_small=cmc_s, mAP_small=np.float64(mAP_s))


# --------------------------------------------------------------------------------------------
# (3) re_ranking, (5) R1_mAP_eval end to end
# --------------------------------------------------------------------------------------------
def min_topk_gap(feat, k):
    """smallest gap between consecutive entries of the top-k of each row of the normalised
    distance matrix (tie-freeness check for the goldens)."""
    f = torch.from_numpy(feat)
    d = (f * f).sum(1, keepdim=True) + (f * f).sum(1, keepdim=True).t() - 2 * f @ f.t()
    d = d.numpy()
    o = (d / d.max(axis=0)).T
    s = np.sort(o, axis=1)[:, :k]
    return float(np.min(np.diff(s, axis=1)))


def gen_rerank():
    N, nq, D = 480, 96, 256
    # choose, among a bounded set of seeds, the one whose top-53 neighbour lists are best separated
    # (the reference's argsort is unstable and MKL's fp32 GEMM differs from ours at the 1e-7 level,
    # so near-ties in the ranking would make the fixture ambiguous)
    best = (-1.0, None)
    for s in range(21, 61):
        feat, pid = synth.clustered_features(N, D, sigma=2.2, seed=s, per_id=8)
        gap = min_topk_gap(feat, 53)
        if gap > best[0]:
            best = (gap, s)
    gap, seed = best
    feat, pid = synth.clustered_features(N, D, sigma=2.2, seed=seed, per_id=8)
    print(f"rerank fixture: seed={seed} min top-53 gap={gap:.3e}")
    q, g = torch.from_numpy(feat[:nq]), torch.from_numpy(feat[nq:])
    out = {"feat": feat, "pid": pid, "nq": nq, "seed": seed}
    cases = [(50, 15, 0.3), (20, 6, 0.3), (5, 1, 0.3), (20, 6, 0.0), (20, 6, 1.0), (7, 3, 0.5), (10, 1, 0.3)]
    for k1, k2, lam in cases:
        r = ref_reranking.re_ranking(q, g, k1, k2, lam)
        assert r.dtype == np.float32 and r.shape == (nq, N - nq)
        out[f"rr_{k1}_{k2}_{lam}"] = r
    out["cases"] = np.array(cases, dtype=np.float64)
    # local_distmat variants (a seeded, NON-symmetric positive N x N matrix)
    rng = np.random.default_rng(5)
    local = (rng.random((N, N)).astype(np.float32) * 0.5).astype(np.float32)
    out["local"] = local.astype(np.float16)   # stored compactly; exactly representable
    local = out["local"].astype(np.float32)
    out["rr_local_20_6_0.3"] = ref_reranking.re_ranking(q, g, 20, 6, 0.3, local_distmat=local)
    out["rr_onlylocal_20_6_0.3"] = ref_reranking.re_ranking(q, g, 20, 6, 0.3, local_distmat=local.copy(),
                                                         only_local=True)
    save("rerank.npz", **out)

    # R1_mAP_eval end to end on the same features (raw = before normalisation)
    raw, pid2 = synth.clustered_features(N, D, sigma=2.2, seed=seed, per_id=8, normalize=False)
    assert np.array_equal(pid, pid2)
    cam = synth.labels_for(N)
    e2e = {"raw": raw.astype(np.float32), "pid": pid, "cam": cam, "nq": nq}
    for rr in (False, True):
        for fn in (True, False):
            ev = ref_metrics.R1_mAP_eval(nq, max_rank=50, feat_norm=fn, reranking=rr)
            ev.reset()
            B = 64
            for s in range(0, N, B):
                ev.update((torch.from_numpy(raw[s:s + B]), tuple(int(x) for x in pid[s:s + B]),
                           tuple(int(x) for x in cam[s:s + B])))
            cmc, mAP, distmat, pids, camids, qf, gf = quiet(ev.compute)
            tag = f"rr{int(rr)}_fn{int(fn)}"
            e2e[f"cmc_{tag}"] = cmc
            e2e[f"mAP_{tag}"] = np.float64(mAP)
            e2e[f"distmat_{tag}"] = distmat.astype(np.float32)
    save("r1_map_eval.npz", **e2e)


def gen_r1_map_eval_edge():
    """R1_mAP_eval's quirks as the reference itself behaves (utils/metrics.py:91-134): `feat_norm` is used as a TRUTH VALUE
    (:112) -- the shipped config passes the string 'yes' (config/defaults_base.py:178), so 'no' normalises too and only a
    falsy value ('' / 0 / False) does not; `max_rank` is stored (:95) but never forwarded to eval_func (:132), so
    max_rank=10 still returns 50 ranks."""
    seed = int(np.load(os.path.join(HERE, "rerank.npz"))["seed"])
    N, nq, D = 480, 96, 256
    raw, pid = synth.clustered_features(N, D, sigma=2.2, seed=seed, per_id=8, normalize=False)
    cam = synth.labels_for(N)
    out = {"seed": seed, "nq": nq}
    cases = {"fn_empty": dict(feat_norm='', reranking=False), "fn_no": dict(feat_norm='no', reranking=False),
             "fn_zero_rr": dict(feat_norm=0, reranking=True), "max_rank_10": dict(max_rank=10, feat_norm='yes', reranking=False),
             "max_rank_10_rr": dict(max_rank=10, feat_norm='yes', reranking=True)}
    for tag, kw in cases.items():
        ev = ref_metrics.R1_mAP_eval(nq, **kw)
        ev.reset()
        for s in range(0, N, 64):
            ev.update((torch.from_numpy(raw[s:s + 64]), tuple(int(x) for x in pid[s:s + 64]), tuple(int(x) for x in cam[s:s + 64])))
        cmc, mAP, distmat, pids, camids, qf, gf = quiet(ev.compute)
        out[f"cmc_{tag}"] = cmc
        out[f"mAP_{tag}"] = np.float64(mAP)
        # the matrices equal those of the four plain cases already committed (r1_map_eval.npz), bit for bit: store which
        twin = {"fn_empty": "rr0_fn0", "fn_no": "rr0_fn1", "fn_zero_rr": "rr1_fn0", "max_rank_10": "rr0_fn1", "max_rank_10_rr": "rr1_fn1"}[tag]
        base = np.load(os.path.join(HERE, "r1_map_eval.npz"))
        assert np.array_equal(distmat.astype(np.float32), base[f"distmat_{twin}"]), tag
        assert np.array_equal(cmc, base[f"cmc_{twin}"]) and float(mAP) == float(base[f"mAP_{twin}"]), tag
        out[f"twin_{tag}"] = np.array(twin)
        out[f"qf_rownorm_{tag}"] = np.linalg.norm(np.asarray(qf), axis=1).astype(np.float32)   # 1.0 iff the features were normalised
        print(tag, "cmc len", len(cmc), "mAP", float(mAP), "|qf| in", float(out[f"qf_rownorm_{tag}"].min()), float(out[f"qf_rownorm_{tag}"].max()))
    save("r1_map_eval_edge.npz", **out)


def gen_rerank_small():
    """N smaller than k1+1 / k2: numpy slicing clamps the neighbour lists and np.mean divides by the clamped
    row count (utils/reranking.py:53-54,60-62,76) - behaviour the build has to follow, pinned here."""
    out = {}
    for tag, (N, nq, D, seed) in {"a": (30, 6, 32, 3), "b": (12, 3, 16, 4), "c": (52, 10, 24, 5)}.items():
        feat, pid = synth.clustered_features(N, D, sigma=1.5, seed=seed, per_id=5)
        q, g = torch.from_numpy(feat[:nq]), torch.from_numpy(feat[nq:])
        out[f"feat_{tag}"] = feat
        out[f"nq_{tag}"] = nq
        for k1, k2, lam in [(50, 15, 0.3), (20, 6, 0.3), (60, 40, 0.5)]:
            out[f"rr_{tag}_{k1}_{k2}_{lam}"] = ref_reranking.re_ranking(q, g, k1, k2, lam)
    save("rerank_small.npz", **out)


# --------------------------------------------------------------------------------------------
# (3b) re_ranking on UNSELECTED seeds at larger N (round-2: the seed of gen_rerank() is chosen for well separated
# neighbour lists; these are not), and np.exp itself
# --------------------------------------------------------------------------------------------
SEED_CASES = [  # (seed, N, D, sigma, per_id, k1, k2, lambda) -- seeds are consecutive integers, nothing is scanned
    (101, 1500, 512, 2.6, 10, 50, 15, 0.3),
    (102, 1000, 256, 2.2, 8, 20, 6, 0.3),
    (103, 2000, 768, 3.0, 20, 50, 15, 0.3),
    (104, 1000, 1280, 3.5, 20, 50, 15, 0.3),
    (105, 3000, 1280, 3.5, 20, 50, 15, 0.3),
    (106, 4000, 768, 3.0, 20, 20, 6, 0.3),
    (107, 2500, 512, 2.6, 10, 50, 15, 0.5),
    (108, 1800, 768, 3.0, 12, 30, 10, 0.3),
    (109, 2000, 1280, 5.5, 20, 50, 15, 0.3),   # noisier: mAP well below 1 so that rank errors would show
    (110, 1500, 768, 4.5, 20, 20, 6, 0.3),
]
N_SAMPLES = 16384


def seed_case_inputs(seed, N, D, sigma, per_id):
    """features of a seed case: raw clustered features (element-wise fp32 arithmetic only, so the same bytes on any
    host) normalised by the ORACLE's l2_normalize (fixed summation order) -- the tests regenerate exactly this."""
    sys.path.insert(0, ROOT)
    from oracle import oracle as orc
    raw, pid = synth.clustered_features(N, D, sigma, seed=seed, per_id=per_id, normalize=False)
    return orc.l2_normalize(raw), pid


def gen_rerank_seeds():
    import hashlib
    sys.path.insert(0, ROOT)
    from oracle import oracle as orc
    out = {"cases": np.array(SEED_CASES, dtype=np.float64), "n_samples": np.int64(N_SAMPLES)}
    for (seed, N, D, sigma, per_id, k1, k2, lam) in SEED_CASES:
        nq = N // 5
        feat, pid = seed_case_inputs(seed, N, D, sigma, per_id)
        q, g = torch.from_numpy(feat[:nq]), torch.from_numpy(feat[nq:])
        cam = synth.labels_for(N)
        tag = f"s{seed}"
        out[f"{tag}_feat_sha"] = np.array(hashlib.sha256(feat.tobytes()).hexdigest())
        # (i) the reference as it is called (its own MKL distance GEMM)
        r = ref_reranking.re_ranking(q, g, k1, k2, lam)
        cmc, mAP = quiet(ref_metrics.eval_func, r, pid[:nq], pid[nq:], cam[:nq], cam[nq:])
        rng = np.random.default_rng(seed)
        flat = rng.choice(r.size, size=N_SAMPLES, replace=False).astype(np.int64)
        out[f"{tag}_idx"] = flat.astype(np.int32)
        out[f"{tag}_val"] = r.reshape(-1)[flat]
        out[f"{tag}_mAP"] = np.float64(mAP)
        out[f"{tag}_cmc"] = cmc
        # (ii) both sides fed the SAME distance matrix (the oracle's k-ascending fmaf-chain GEMM) through
        # local_distmat / only_local=True: everything after the GEMM is then comparable bit for bit
        d_or = orc.euclidean_distance(feat, feat)
        r2 = ref_reranking.re_ranking(q, g, k1, k2, lam, local_distmat=d_or.copy(), only_local=True)
        out[f"{tag}_sameD_sha"] = np.array(hashlib.sha256(np.ascontiguousarray(r2).tobytes()).hexdigest())
        out[f"{tag}_sameD_val"] = r2.reshape(-1)[flat]
        cmc2, mAP2 = quiet(ref_metrics.eval_func, r2, pid[:nq], pid[nq:], cam[:nq], cam[nq:])
        out[f"{tag}_sameD_mAP"] = np.float64(mAP2)
        # measured oracle-vs-reference deviation over the FULL matrices (reported in DESIGN.md section 2)
        o1 = orc.re_ranking(feat[:nq], feat[nq:], k1, k2, lam)
        o2 = orc.re_ranking(feat[:nq], feat[nq:], k1, k2, lam, local_distmat=d_or, only_local=True)
        d1, d2 = np.abs(o1 - r), np.abs(o2 - r2)
        cmc_o, mAP_o = orc.eval_func(o1, pid[:nq], pid[nq:])
        out[f"{tag}_measured"] = np.array([(d1 > 1e-5).mean(), d1.max(), (d2 != 0).mean(), d2.max(),
                                           abs(mAP_o - mAP), np.abs(cmc_o - cmc).max()], dtype=np.float64)
        print(f"{tag}: N={N} D={D} k=({k1},{k2}) lam={lam} mAP_ref={mAP:.4f}  as-called: frac>1e-5 {(d1 > 1e-5).mean():.2e} "
              f"max {d1.max():.2e} dmAP {abs(mAP_o - mAP):.1e} | same-D: differing entries {(d2 != 0).mean():.2e} "
              f"max {d2.max():.2e} bit-equal {np.array_equal(o2, r2)}")
    save("rerank_seeds.npz", **out)


def gen_np_exp():
    """np.exp(float32) of this image's numpy (the reference calls it at utils/reranking.py:70)"""
    rng = np.random.default_rng(3)
    x = np.concatenate([-rng.random(30000), rng.uniform(-104.5, 89.0, 6000), rng.standard_normal(4000) * 1e-3,
                        np.array([0.0, -0.0, -1.0, 1.0, -87.5, -103.9, -103.98, 88.72, 88.73, -1e-8])]).astype(np.float32)
    with np.errstate(over="ignore"):
        y = np.exp(x)
    assert y.dtype == np.float32
    save("np_exp.npz", x=x, y=y, numpy_version=np.array(np.__version__))


# --------------------------------------------------------------------------------------------
# (4) VisionTransformer
# --------------------------------------------------------------------------------------------
def run_vit(cfg, sd_np, imgs, cv=None):
    m = ref_clip.VisionTransformer(cfg["h_res"], cfg["w_res"], cfg["patch"], cfg["stride"], cfg["width"],
                                   cfg["layers"], cfg["heads"], cfg["out_dim"])
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    m.eval()
    with torch.no_grad():
        x11, x12, xproj = m(torch.from_numpy(imgs), None if cv is None else torch.from_numpy(cv))
    return torch.cat([x12[:, 0], xproj[:, 0]], dim=1).numpy(), x12.numpy()


def gen_vit():
    out = {}
    # (i) reduced: width 128, 2 layers, 2 heads (d_h 64), 64x32 input, stride 16 -> 4x2 grid, L = 9
    small = dict(h_res=4, w_res=2, patch=16, stride=16, width=128, layers=2, heads=2, out_dim=64)
    imgs = synth.synthetic_images(3, 64, 32, seed=3)
    sd = synth.vit_state_dict(small, seed=7, std=0.05, ln_jitter=0.1)
    f, x12 = run_vit(small, sd, imgs)
    out["small_feat"] = f
    out["small_x12"] = x12
    # (ii) full ViT-B/16 (256x128 input, L = 129), 4 images, weights regenerated from the seed
    big = synth.VIT_B16
    imgs = synth.synthetic_images(4, 256, 128, seed=1234)
    sd = synth.vit_state_dict(big, seed=7, std=0.02, ln_jitter=0.05)
    f, _ = run_vit(big, sd, imgs)
    out["b16_feat"] = f
    # (iii-a) with a camera/view embedding added to the CLS token
    cv = (np.random.default_rng(17).standard_normal((4, 768)) * 0.02 * 3.0).astype(np.float32)
    f, _ = run_vit(big, sd, imgs, cv)
    out["b16_cv"] = cv
    out["b16_feat_cv"] = f
    # (iii-b) stride 12 -> 21 x 10 grid, L = 211 (overlapping patches)
    s12 = dict(big, h_res=21, w_res=10, stride=12)
    sd12 = synth.vit_state_dict(s12, seed=8, std=0.02, ln_jitter=0.05)
    f, _ = run_vit(s12, sd12, imgs[:2])
    out["b16_s12_feat"] = f
    save("vit.npz", **out)


def gen_head():
    """build_transformer.forward, eval branch (model/make_model.py:89-115), on the reference's own VisionTransformer class.
    model/make_model.py itself cannot be imported here (torchvision / timm / the CLIP download), so its
    head is composed from the pieces it is made of: the SIE index rules :89-96 (three one-line expressions, restated
    below), torch.nn.BatchNorm1d in eval mode (:58-63, :102-103: torch's published operator, here with NON-trivial running
    statistics and affine parameters) and the two torch.cat of :110-115."""
    big = synth.VIT_B16
    sd = synth.vit_state_dict(big, seed=21, std=0.02, ln_jitter=0.05)
    imgs = synth.synthetic_images(6, 256, 128, seed=77)
    m = ref_clip.VisionTransformer(big["h_res"], big["w_res"], big["patch"], big["stride"], big["width"], big["layers"],
                                   big["heads"], big["out_dim"])
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m.eval()
    camera_num, view_num, sie_coe = 6, 3, 3.0
    rng = np.random.default_rng(5)
    cam = rng.integers(0, camera_num, 6).astype(np.int64)
    view = rng.integers(0, view_num, 6).astype(np.int64)
    out = {"cam": cam, "view": view, "camera_num": camera_num, "view_num": view_num, "sie_coe": sie_coe}
    bns = {}
    for name, n in (("bottleneck", 768), ("bottleneck_proj", 512)):
        bn = torch.nn.BatchNorm1d(n)
        with torch.no_grad():
            bn.weight.copy_(torch.from_numpy((1 + 0.2 * rng.standard_normal(n)).astype(np.float32)))
            bn.bias.copy_(torch.from_numpy((0.1 * rng.standard_normal(n)).astype(np.float32)))
            bn.running_mean.copy_(torch.from_numpy((0.3 * rng.standard_normal(n)).astype(np.float32)))
            bn.running_var.copy_(torch.from_numpy((0.4 + rng.random(n)).astype(np.float32)))
        bn.eval()
        bns[name] = bn
        for k in ("weight", "bias", "running_mean", "running_var"):
            out[f"{name}.{k}"] = getattr(bn, k).detach().numpy().copy()
    rows = {"cam_view": camera_num * view_num, "cam": camera_num, "view": view_num}
    with torch.no_grad():
        for mode in ("cam_view", "cam", "view", "none"):
            cv = None
            if mode != "none":
                table = torch.from_numpy((0.02 * rng.standard_normal((rows[mode], 768))).astype(np.float32))
                out[f"cv_embed_{mode}"] = table.numpy()
                cam_label, view_label = torch.from_numpy(cam), torch.from_numpy(view)
                if mode == "cam_view":      # model/make_model.py:89-90
                    cv = sie_coe * table[cam_label * view_num + view_label]
                elif mode == "cam":         # :91-92
                    cv = sie_coe * table[cam_label]
                else:                       # :93-94
                    cv = sie_coe * table[view_label]
            _, x12, xproj = m(torch.from_numpy(imgs), cv)
            img_feature, img_feature_proj = x12[:, 0], xproj[:, 0]                     # :98-100
            feat, feat_proj = bns["bottleneck"](img_feature), bns["bottleneck_proj"](img_feature_proj)   # :102-103
            out[f"{mode}_after"] = torch.cat([feat, feat_proj], dim=1).numpy()          # :110-112
            out[f"{mode}_before"] = torch.cat([img_feature, img_feature_proj], dim=1).numpy()   # :113-115
    save("head.npz", **out)


def gen_resize():
    """T.Resize of val_transforms (datasets/make_dataloader.py:57-58): torchvision hands the PIL image to
    Image.resize(size[::-1], BILINEAR).  torchvision is not installed here; the Pillow call it makes is.  Ragged
    uint8 inputs -> (out_h, out_w) outputs; a TTA golden for the four views of
    processor/processor_uniprompt_stage2.py:605-633 rides along (tensor ops only, computed with torch)."""
    from PIL import Image
    import PIL
    rng = np.random.default_rng(77)
    cases = [((128, 64), (256, 128)),    # Market-1501 native size -> SIZE_TEST, 2x up
             ((64, 32), (64, 32)),       # already the target size (Image.resize returns a copy)
             ((301, 97), (256, 128)),    # down in h, up in w, odd sizes
             ((259, 107), (128, 48)),    # > 2x down (wider filter support: 5 taps)
             ((40, 33), (64, 32)),       # small, ~1x in w
             ((1, 1), (8, 4)),           # degenerate
             ((200, 50), (32, 16))]      # 6.25x / 3.1x down
    out = {"n": np.int64(len(cases)), "pillow_version": np.array(PIL.__version__)}
    for i, ((h, w), (oh, ow)) in enumerate(cases):
        # smooth-ish + noise so that clipping at 0/255 and all rounding branches occur
        base = rng.integers(0, 256, ((h + 7) // 8, (w + 7) // 8, 3)).astype(np.float32)
        img = np.kron(base, np.ones((8, 8, 1), np.float32))[:h, :w]
        img = np.clip(img + rng.normal(0, 40, img.shape), 0, 255).astype(np.uint8)
        res = np.asarray(Image.fromarray(img, "RGB").resize((ow, oh), Image.BILINEAR))
        out[f"in{i}"] = img
        out[f"out{i}"] = res
    save("resize.npz", **out)


def gen_tta():
    """TTA "option A" of the Uni-Prompt evaluation (processor/processor_uniprompt_stage2.py:598-640): the four views
    are built with the reference's own tensor expressions, encoded by the reference VisionTransformer, averaged
    with torch.stack(...).mean(0) and normalised with F.normalize."""
    import torch.nn.functional as F
    out = {}
    small = dict(h_res=4, w_res=2, patch=16, stride=16, width=128, layers=2, heads=2, out_dim=64)
    sd = synth.vit_state_dict(small, seed=7, std=0.05, ln_jitter=0.1)
    for tag, imgs in (("f32", synth.synthetic_images(5, 64, 32, seed=21)),):
        img = torch.from_numpy(imgs)
        views = [img, torch.flip(img, [3]), img.mean(dim=1, keepdim=True).repeat(1, 3, 1, 1),
                 img[:, 0:1, :, :].repeat(1, 3, 1, 1)]
        feats = [torch.from_numpy(run_vit(small, sd, v.contiguous().numpy())[0]) for v in views]
        out[f"{tag}_views"] = torch.stack(feats, 0).numpy()
        agg = torch.stack(feats, dim=0).mean(dim=0)
        out[f"{tag}_mean"] = agg.numpy()
        out[f"{tag}_mean_norm"] = F.normalize(agg, p=2, dim=1).numpy()
    # uint8 HWC input through ToTensor + Normalize(mean, std) first (datasets/make_dataloader.py:59-60)
    rng = np.random.default_rng(5)
    u8 = rng.integers(0, 256, (4, 64, 32, 3), dtype=np.uint8)
    mean, std = np.array([0.5, 0.4, 0.45], np.float32), np.array([0.5, 0.25, 0.3], np.float32)
    t = torch.from_numpy(u8).permute(0, 3, 1, 2).to(torch.float32).div(255)
    t = (t - torch.from_numpy(mean)[None, :, None, None]) / torch.from_numpy(std)[None, :, None, None]
    views = [t, torch.flip(t, [3]), t.mean(dim=1, keepdim=True).repeat(1, 3, 1, 1), t[:, 0:1].repeat(1, 3, 1, 1)]
    feats = [torch.from_numpy(run_vit(small, sd, v.contiguous().numpy())[0]) for v in views]
    out["u8_img"], out["u8_mean"], out["u8_std"] = u8, mean, std
    out["u8_views"] = torch.stack(feats, 0).numpy()
    out["u8_mean_norm"] = F.normalize(torch.stack(feats, 0).mean(0), p=2, dim=1).numpy()
    save("tta.npz", **out)


def run_rn50(cfg, sd_np, imgs):
    """reference ModifiedResNet (model/clip/model.py:92-148) + the RN50 eval branch of build_transformer.forward
    (model/make_model.py:82-86, 113-115, NECK_FEAT 'before'): cat(avg_pool(x4), attnpool(x4)[0])"""
    import torch.nn.functional as F
    m = ref_clip.ModifiedResNet(layers=cfg["layers"], output_dim=cfg["out_dim"], heads=cfg["heads"],
                                input_resolution=cfg["h_res"] * cfg["w_res"], width=cfg["width"])
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()})
    m.eval()
    with torch.no_grad():
        x = torch.from_numpy(imgs)
        x3, x4, xproj = m(x)
        feat = F.avg_pool2d(x4, x4.shape[2:4]).view(x.shape[0], -1)
    return torch.cat([feat, xproj[0]], dim=1).numpy(), x3.numpy(), x4.numpy()


def gen_rn50():
    out = {}
    # (i) reduced: width 16, layers (1,2,1,1), 64x32 input -> 4x2 final grid, embed 512, 8 heads
    small = dict(layers=(1, 2, 1, 1), width=16, heads=8, out_dim=64, h_res=4, w_res=2)
    imgs = synth.synthetic_images(3, 64, 32, seed=31)
    f, x3, x4 = run_rn50(small, synth.rn50_state_dict(small, seed=11), imgs)
    out["small_feat"], out["small_x4"] = f, x4
    # (ii) full RN50, 256x128 input, 3 images
    imgs = synth.synthetic_images(3, 256, 128, seed=32)
    f, x3, x4 = run_rn50(synth.RN50, synth.rn50_state_dict(synth.RN50, seed=11), imgs)
    out["rn50_feat"] = f
    out["rn50_x4_mean"] = x4.mean(axis=(2, 3))
    save("rn50.npz", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["distance", "rerank", "vit"]
    if "distance" in which:
        gen_distance()
    if "rerank" in which:
        gen_rerank()
    if "r1_edge" in which:
        gen_r1_map_eval_edge()
    if "rerank_small" in which:
        gen_rerank_small()
    if "rerank_seeds" in which:
        gen_rerank_seeds()
    if "np_exp" in which:
        gen_np_exp()
    if "vit" in which:
        gen_vit()
    if "head" in which:
        gen_head()
    if "resize" in which:
        gen_resize()
    if "tta" in which:
        gen_tta()
    if "rn50" in which:
        gen_rn50()
