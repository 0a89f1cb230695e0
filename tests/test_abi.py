"""CPU-side checks of the C ABI: the library loads, exports every symbol include/mpreid.h declares,
and the product path fails loudly (no CPU fallback) when there is no GPU."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from mpreid import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "mpreid.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mpreid_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = _lib.load()
    declared = _declared()
    assert declared == sorted(_lib.SYMBOLS), (declared, sorted(_lib.SYMBOLS))
    for name in declared:
        assert hasattr(L, name), name
    assert L.mpreid_version() >= 100


def test_workspace_queries_need_no_gpu():
    L = _lib.load()
    assert L.mpreid_distance_workspace_bytes(100, 200, 64, 0) >= 300 * 4
    n = 19281
    b = L.mpreid_rerank_workspace_bytes(3368, 15913, 1280, 50, 15, 0)
    assert b > 4 * n * n
    cfg = _lib.VitCfg(256, 128, 16, 16, 16, 8, 768, 12, 12, 512, 0, 0)
    assert L.mpreid_vit_workspace_bytes(ctypes.byref(cfg), 64) > 64 * 129 * 768 * 4


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_product_path_fails_loudly_without_gpu():
    from utils import metrics, reranking
    q = torch.randn(4, 8)
    with pytest.raises(RuntimeError):
        metrics.euclidean_distance(q, q)
    with pytest.raises(RuntimeError):
        reranking.re_ranking(q, q, 2, 1, 0.3)
    ev = metrics.R1_mAP_eval(2)
    ev.reset()
    with pytest.raises(RuntimeError):
        ev.update((q, (0, 1, 0, 1), (0, 0, 1, 1)))


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under mp-reid_amd/ may reference it."""
    bad = []
    pkg = os.path.join(ROOT, "mp-reid_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "mpreid_oracle" in txt.replace(
                        "oracle/mpreid_oracle.c", ""):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_dropin_modules_import_and_dataloader_contract_without_gpu():
    """host logic of the drop-in modules (reference test.py:2-7, test_uniprompt.py:2-7): imports, config keys, the
    val tuple, query-then-gallery order, decoded-image batches"""
    import numpy as np
    from config import cfg, cfg_base
    from datasets.make_dataloader import make_dataloader, RawImageBatch, raw_val_collate_fn
    from datasets import make_dataloader_uniprompt
    import processor.processor as p1
    import processor.processor_uniprompt_stage2 as p2
    import model.make_model_uniprompt as mu
    assert callable(p1.do_inference) and callable(p2.do_inference) and callable(p2.do_inference_ttpt_option_a)
    assert callable(mu.make_model) and callable(make_dataloader_uniprompt.make_dataloader)
    assert cfg.TEST.get('TTA_ENABLED', True) is False and cfg.TEST.TTPT.STEPS == 5 and cfg_base.TEST.FEAT_NORM == "yes"
    c = cfg.clone()
    c.defrost()
    c.merge_from_list(["DATASETS.SYNTH_QUERY", 5, "DATASETS.SYNTH_GALLERY", 9, "TEST.IMS_PER_BATCH", 4,
                       "DATASETS.SYNTH_RAW", True])
    tl, tn, val_loader, num_query, num_classes, cam_num, view_num = make_dataloader(c)
    assert tl is None and tn is None and num_query == 5 and len(val_loader) == 4
    n = 0
    for img, pids, camids, camids_t, views_t, paths in val_loader:
        assert isinstance(img, RawImageBatch) and img.to("cuda") is img
        assert all(a.dtype == np.uint8 and a.ndim == 3 and a.shape[2] == 3 for a in img)
        assert len(pids) == len(img) == camids_t.shape[0] == views_t.shape[0] == len(paths)
        n += len(img)
    assert n == 14
    b = raw_val_collate_fn([(np.zeros((3, 2, 3), np.uint8), 1, 2, 0, "a.jpg"), (np.zeros((5, 4, 3), np.uint8), 3, 4, 0, "b.jpg")])
    assert isinstance(b[0], RawImageBatch) and b[1] == (1, 3) and b[3].tolist() == [2, 4] and b[5] == ("a.jpg", "b.jpg")
    with pytest.raises(NotImplementedError):
        p2.do_inference_ttpt_clipstyle(c, None, None, 0)


def test_fold_conv_bn_matches_conv_plus_batchnorm_on_cpu():
    """host-side BatchNorm folding of the RN50 path (mpreid.ops.fold_conv_bn): conv'(x) + b' == BN(conv(x)) in fp32,
    k order (kh, kw, c), output rows padded to 128, input channels padded on request"""
    import torch.nn.functional as F
    from mpreid import ops
    g = torch.Generator().manual_seed(5)
    for cout, cin, k, cpad in ((24, 16, 3, 64), (64, 32, 1, None), (130, 8, 3, 8)):
        w = torch.randn((cout, cin, k, k), generator=g) * 0.1
        bn = (1 + 0.1 * torch.randn(cout, generator=g), 0.1 * torch.randn(cout, generator=g),
              0.1 * torch.randn(cout, generator=g), 0.5 + torch.rand(cout, generator=g))
        x = torch.randn((2, cin, 6, 5), generator=g)
        ref = F.batch_norm(F.conv2d(x, w, None, padding=k // 2), bn[2], bn[3], bn[0], bn[1], training=False, eps=1e-5)
        wk, bk = ops.fold_conv_bn(w.numpy(), tuple(t.numpy() for t in bn), cin_pad=cpad)
        cp = cpad or cin
        assert wk.dtype == torch.float16 and wk.shape == ((cout + 127) // 128 * 128, k * k * cp) and bk.shape[0] == wk.shape[0]
        assert float(wk[cout:].abs().max()) == 0.0 if wk.shape[0] > cout else True
        w4 = wk[:cout].float().reshape(cout, k, k, cp)[..., :cin].permute(0, 3, 1, 2).contiguous()
        got = F.conv2d(x, w4, bk[:cout], padding=k // 2)
        assert torch.allclose(got, ref, rtol=2e-3, atol=2e-3)          # fp16 rounding of the folded weights only
        if cp > cin:
            assert float(wk[:cout].float().reshape(cout, k, k, cp)[..., cin:].abs().max()) == 0.0


def test_do_inference_batch_grouping_helpers():
    """processor.grouped_batches / merge_batches: consecutive loader batches up to the target, order kept, the
    evaluator still gets the loader's own batches (reference processor/processor.py:187-198 semantics)"""
    from processor.processor import grouped_batches, merge_batches
    from datasets.make_dataloader import RawImageBatch
    batches = []
    for i, n in enumerate((3, 4, 2, 5, 1)):
        batches.append((torch.full((n, 3, 2, 2), float(i)), tuple(range(n)), tuple([i] * n),
                        torch.full((n,), i, dtype=torch.int64), torch.zeros(n, dtype=torch.int64), tuple(f"{i}_{j}" for j in range(n))))
    groups = list(grouped_batches(batches, 6))
    assert [[len(b[1]) for b in g] for g in groups] == [[3, 4], [2, 5], [1]]
    img, cam, view = merge_batches(groups[0], "cpu")
    assert img.shape == (7, 3, 2, 2) and cam.tolist() == [0, 0, 0, 1, 1, 1, 1] and view.shape == (7,)
    assert float(img[2, 0, 0, 0]) == 0.0 and float(img[3, 0, 0, 0]) == 1.0
    raw = [(RawImageBatch([np.zeros((2, 2, 3), np.uint8)] * n), tuple(range(n)), tuple([0] * n),
            torch.zeros(n, dtype=torch.int64), torch.zeros(n, dtype=torch.int64), ("p",) * n) for n in (2, 3)]
    img, cam, view = merge_batches(raw, "cpu")
    assert isinstance(img, RawImageBatch) and len(img) == 5


def test_shipped_eval_configs_load():
    """every YAML under mp-reid_amd/configs merges into the default config (test.py's --config_file path) and names a model
    the build has an encoder for"""
    import glob
    from config import cfg_base
    files = sorted(glob.glob(os.path.join(ROOT, "mp-reid_amd", "configs", "*", "*.yml")))
    assert len(files) >= 3
    for f in files:
        c = cfg_base.clone()
        c.merge_from_file(f)
        assert c.MODEL.NAME in ("ViT-B-16", "RN50") and c.TEST.EVAL is True, f
        assert c.MODEL.ENCODER_PRECISION in ("split", "fp16", "fp32"), f
        assert list(c.INPUT.SIZE_TEST) == [256, 128] and c.TEST.FEAT_NORM == "yes", f


def test_ablation_build_is_refused_unless_asked_for(tmp_path):
    """MPREID_LIB may point at any build; one that says it is a timing-ablation build (-DMPREID_ABLATION: wrong results by
    design) is refused by mpreid._lib.load() unless MPREID_ALLOW_ABLATION=1, a stale build without the export too; the
    product library answers 0 and lives alone in mpreid/ (ablation builds go to tools/ablation_lib/)."""
    import subprocess
    import sys
    assert _lib.load().mpreid_is_ablation_build() == 0
    pkg = os.path.join(ROOT, "mp-reid_amd", "mpreid")
    assert [f for f in os.listdir(pkg) if f.endswith(".so")] == ["libmpreid_hip.so"]
    src = tmp_path / "fake.c"
    src.write_text("int mpreid_is_ablation_build(void) { return 1; }\nint mpreid_version(void) { return 100; }\n")
    fake = tmp_path / "libfake_abl.so"
    subprocess.run(["gcc", "-shared", "-fPIC", "-o", str(fake), str(src)], check=True)
    stale = tmp_path / "libfake_stale.so"
    (tmp_path / "stale.c").write_text("int mpreid_version(void) { return 100; }\n")
    subprocess.run(["gcc", "-shared", "-fPIC", "-o", str(stale), str(tmp_path / "stale.c")], check=True)
    code = ("import sys; sys.path.insert(0, %r)\nfrom mpreid import _lib\ntry:\n    _lib.load()\nexcept RuntimeError as e:\n"
            "    print('REFUSED', e)\nexcept AttributeError as e:\n    print('LOADED-PAST-GUARD')\n" % os.path.join(ROOT, "mp-reid_amd"))
    env = {k: v for k, v in os.environ.items() if k != "MPREID_ALLOW_ABLATION"}
    r = subprocess.run([sys.executable, "-c", code], env=dict(env, MPREID_LIB=str(fake)), capture_output=True, text=True)
    assert "REFUSED" in r.stdout and "timing-ablation build" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([sys.executable, "-c", code], env=dict(env, MPREID_LIB=str(stale)), capture_output=True, text=True)
    assert "REFUSED" in r.stdout and "stale build" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([sys.executable, "-c", code], env=dict(env, MPREID_LIB=str(fake), MPREID_ALLOW_ABLATION="1"),
                       capture_output=True, text=True)
    assert "LOADED-PAST-GUARD" in r.stdout, r.stdout + r.stderr     # (the fake exports nothing else: the prototypes fail)
