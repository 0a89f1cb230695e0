"""ctypes binding of libmpreid_hip.so (include/mpreid.h).

There is no CPU fallback: if the library is missing, or a compute entry point is called without a
GPU, a RuntimeError is raised (SURVEY.md §8b "Errors").
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
#: MPREID_LIB points at another build of the same library (kernel A/B runs); there is no other fallback
LIB_PATH = os.environ.get("MPREID_LIB") or os.path.join(_HERE, "libmpreid_hip.so")

GEMM_F32_EXACT = 0
GEMM_F16_FAST = 1
GEMM_F16_SPLIT3 = 2
RERANK_AUTO, RERANK_DENSE, RERANK_SPARSE, RERANK_SPARSE_SPLIT3 = 0, 1, 2, 3
ERR_RETRY_DENSE = -5

#: every symbol include/mpreid.h declares (tests check the library exports all of them)
SYMBOLS = [
    "mpreid_version", "mpreid_is_ablation_build", "mpreid_last_error", "mpreid_device_count", "mpreid_device_info",
    "mpreid_sqnorm_f32", "mpreid_l2_normalize_f32", "mpreid_distance_workspace_bytes",
    "mpreid_euclidean_distance_f32", "mpreid_cosine_similarity_f32",
    "mpreid_rerank_workspace_bytes", "mpreid_rerank_f32", "mpreid_rerank_debug_copy",
    "mpreid_rerank_workspace_bytes_ex", "mpreid_rerank_f32_ex", "mpreid_rerank_debug_copy_ex",
    "mpreid_eval_rank_positions", "mpreid_rr_dist_rows", "mpreid_rr_vcap", "mpreid_rr_krecip", "mpreid_rr_krecip_scratch_bytes",
    "mpreid_rr_sparse_workspace_bytes", "mpreid_rr_neighbours_sparse", "mpreid_rr_krecip_sparse", "mpreid_rr_pack_rows", "mpreid_rr_rowptr", "mpreid_rr_ell_to_csr", "mpreid_rr_csr_to_ell", "mpreid_rr_qe_count",
    "mpreid_rr_qe_fill", "mpreid_rr_jaccard", "mpreid_rr_jaccard_hist_bytes",
    "mpreid_rr_csc_chunks", "mpreid_rr_csc_count", "mpreid_rr_csc_fill", "mpreid_rr_jaccard_indexed",
    "mpreid_vit_workspace_bytes", "mpreid_vit_forward", "mpreid_vit_forward_u8", "mpreid_vit_forward_view",
    "mpreid_vit_workspace_bytes_f32", "mpreid_vit_forward_f32", "mpreid_vit_forward_f32_view",
    "mpreid_tta_mean_f32", "mpreid_resize_workspace_bytes", "mpreid_resize_bilinear_u8", "mpreid_conv_f16_nhwc",
    "mpreid_rn50_workspace_bytes", "mpreid_rn50_forward", "mpreid_rn50_workspace_bytes_f32", "mpreid_rn50_forward_f32",
    "mpreid_rn50_workspace_bytes_split", "mpreid_rn50_forward_split", "mpreid_rn50_forward_f32_u8", "mpreid_rn50_forward_split_u8",
    "mpreid_rn50_forward_f32_view", "mpreid_rn50_forward_split_view",
    "mpreid_gemm_f16_nt", "mpreid_gemm_f16_nt_ex", "mpreid_gemm_f16_split_nt", "mpreid_split_pack_f32",
    "mpreid_cast_f32_to_f16", "mpreid_profile_enable", "mpreid_profile_reset", "mpreid_profile_query",
]


class RerankStats(C.Structure):
    _fields_ = [("n", C.c_int64), ("k1", C.c_int32), ("k2", C.c_int32), ("half_k1", C.c_int32),
                ("v_cap", C.c_int32), ("vqe_cap", C.c_int32), ("v_nnz", C.c_int64), ("vqe_nnz", C.c_int64),
                ("jaccard_pairs", C.c_int64), ("krecip_r_sum", C.c_int64), ("fallback_rows", C.c_int64),
                ("cand_total", C.c_int64), ("algo", C.c_int32), ("ms_gemm", C.c_float), ("ms_topk", C.c_float),
                ("ms_krecip", C.c_float), ("ms_qe", C.c_float), ("ms_csc", C.c_float),
                ("ms_jaccard", C.c_float), ("ms_total", C.c_float), ("ms_dq", C.c_float)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class VitCfg(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("img_h", "img_w", "patch", "stride", "h_res", "w_res", "width", "layers",
                                         "heads", "out_dim", "neck_after", "cls_only_last", "precision")]


class VitLayer(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("in_proj_w", "in_proj_b", "out_proj_w", "out_proj_b", "ln1_g", "ln1_b",
                                          "ln2_g", "ln2_b", "fc_w", "fc_b", "proj_w", "proj_b")] + \
               [(k, C.c_float) for k in ("in_proj_s", "out_proj_s", "fc_s", "proj_s")] + \
               [(k, C.c_void_p) for k in ("in_proj_c", "fc_c")]


class VitWeights(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("conv_w", "class_emb", "pos_emb", "ln_pre_g", "ln_pre_b", "ln_post_g",
                                          "ln_post_b", "proj", "bn_scale", "bn_shift", "bn_proj_scale",
                                          "bn_proj_shift")] + [("layers", C.POINTER(VitLayer)), ("conv_s", C.c_float)]


class Rn50Conv(C.Structure):
    _fields_ = [("w", C.c_void_p), ("bias", C.c_void_p), ("cin", C.c_int32), ("cout", C.c_int32),
                ("cout_pad", C.c_int32), ("taps", C.c_int32)]


class Rn50Block(C.Structure):
    _fields_ = [("conv1", Rn50Conv), ("conv2", Rn50Conv), ("conv3", Rn50Conv), ("down", Rn50Conv), ("stride", C.c_int32)]


class Rn50Cfg(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("img_h", "img_w", "width", "n_blocks", "heads", "out_dim")]


class Rn50Weights(C.Structure):
    _fields_ = [("stem1_w", C.c_void_p), ("stem1_b", C.c_void_p), ("stem2", Rn50Conv), ("stem3", Rn50Conv),
                ("blocks", C.POINTER(Rn50Block)), ("pos_emb", C.c_void_p), ("kt_w", C.c_void_p), ("v_w", C.c_void_p), ("v_b", C.c_void_p),
                ("q_w", C.c_void_p), ("q_b", C.c_void_p), ("c_w", C.c_void_p), ("c_b", C.c_void_p),
                ("bn_scale", C.c_void_p), ("bn_shift", C.c_void_p)]


class Rn50ConvF32(C.Structure):
    _fields_ = [("w", C.c_void_p), ("bias", C.c_void_p), ("cin", C.c_int32), ("cout", C.c_int32), ("taps", C.c_int32)]


class Rn50BlockF32(C.Structure):
    _fields_ = [("conv1", Rn50ConvF32), ("conv2", Rn50ConvF32), ("conv3", Rn50ConvF32), ("down", Rn50ConvF32),
                ("stride", C.c_int32)]


class Rn50WeightsF32(C.Structure):
    _fields_ = [("stem1_w", C.c_void_p), ("stem1_b", C.c_void_p), ("stem2", Rn50ConvF32), ("stem3", Rn50ConvF32),
                ("blocks", C.POINTER(Rn50BlockF32)), ("pos_emb", C.c_void_p), ("q_w", C.c_void_p), ("q_b", C.c_void_p),
                ("k_w", C.c_void_p), ("k_b", C.c_void_p), ("v_w", C.c_void_p), ("v_b", C.c_void_p), ("c_w", C.c_void_p),
                ("c_b", C.c_void_p), ("bn_scale", C.c_void_p), ("bn_shift", C.c_void_p)]


class Rn50ConvSplit(C.Structure):
    _fields_ = [("w", C.c_void_p), ("bias", C.c_void_p), ("cin", C.c_int32), ("cout", C.c_int32), ("taps", C.c_int32),
                ("kseg", C.c_int32), ("npad", C.c_int32), ("oscale", C.c_float)]


class Rn50BlockSplit(C.Structure):
    _fields_ = [("conv1", Rn50ConvSplit), ("conv2", Rn50ConvSplit), ("conv3", Rn50ConvSplit), ("down", Rn50ConvSplit),
                ("stride", C.c_int32)]


class Rn50WeightsSplit(C.Structure):
    _fields_ = [("f32", Rn50WeightsF32), ("blocks", C.POINTER(Rn50BlockSplit)), ("k", Rn50ConvSplit), ("v", Rn50ConvSplit),
                ("stem2", Rn50ConvSplit), ("stem3", Rn50ConvSplit)]


class ProfileEntry(C.Structure):
    _fields_ = [("epilogue", C.c_int32), ("n", C.c_int32), ("k", C.c_int32), ("m", C.c_int64),
                ("launches", C.c_int64), ("total_ms", C.c_double), ("flops_total", C.c_double)]


GEMM_EPILOGUE_NAMES = {0: "f32", 1: "qkv_bias_f16", 2: "bias_residual", 3: "fc_bias_quickgelu", 4: "patch_embed",
                       5: "euclid", 6: "cosine", 7: "conv1x1_bias_relu", 8: "conv1x1_bias_residual_relu", 9: "candidates",
                       10: "split_qkv_bias_f32", 11: "split_bias_residual", 12: "split_fc_bias_quickgelu",
                       13: "split_patch_embed"}
VIT_F16, VIT_SPLIT = 0, 1

_lib = None


def load():
    """Load the library and declare prototypes.  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64; it must be in the process BEFORE this library is dlopen-ed so that the
    # dynamic linker resolves our dependency to the same runtime.  Loaded the other way round the process holds
    # two HIP runtimes and the second one reports "no ROCm-capable device".
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python mp-reid_amd/mpreid/build.py` "
            "(or __graft_entry__.build()).  There is no CPU fallback for the HIP path.")
    L = C.CDLL(LIB_PATH)
    # MPREID_LIB may point at any build of the library.  One compiled with -DMPREID_ABLATION skips work on request (wrong
    # results by design): it must say so itself, and is refused unless the caller asked for exactly that.
    if not hasattr(L, "mpreid_is_ablation_build"):
        raise RuntimeError(f"{LIB_PATH} does not export mpreid_is_ablation_build: a stale build; rebuild it")
    L.mpreid_is_ablation_build.restype = C.c_int
    if L.mpreid_is_ablation_build() and os.environ.get("MPREID_ALLOW_ABLATION") != "1":
        raise RuntimeError(f"{LIB_PATH} is a timing-ablation build (-DMPREID_ABLATION: wrong results by design); "
                           "set MPREID_ALLOW_ABLATION=1 to load it for a measurement")
    vp, i64, i32, f32, f64, sz = C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_double, C.c_size_t
    L.mpreid_version.restype = i32
    L.mpreid_last_error.restype = C.c_char_p
    L.mpreid_device_count.restype = i32
    L.mpreid_device_info.restype = i32
    L.mpreid_device_info.argtypes = [C.c_char_p, i32, C.POINTER(i32), C.POINTER(sz)]
    L.mpreid_sqnorm_f32.restype = i32
    L.mpreid_sqnorm_f32.argtypes = [vp, i64, i32, vp, vp]
    L.mpreid_l2_normalize_f32.restype = i32
    L.mpreid_l2_normalize_f32.argtypes = [vp, i64, i32, f32, vp, vp]
    L.mpreid_distance_workspace_bytes.restype = sz
    L.mpreid_distance_workspace_bytes.argtypes = [i64, i64, i32, i32]
    for f in (L.mpreid_euclidean_distance_f32, L.mpreid_cosine_similarity_f32):
        f.restype = i32
        f.argtypes = [vp, vp, i64, i64, i32, vp, i64, i32, vp, sz, vp]
    L.mpreid_rerank_workspace_bytes.restype = sz
    L.mpreid_rerank_workspace_bytes.argtypes = [i64, i64, i32, i32, i32, i32]
    L.mpreid_rerank_f32.restype = i32
    L.mpreid_rerank_f32.argtypes = [vp, vp, i64, i64, i32, i32, i32, f64, vp, i32, vp, i64, vp, sz, vp,
                                    C.POINTER(RerankStats), i32]
    L.mpreid_rerank_debug_copy.restype = i32
    L.mpreid_rerank_debug_copy.argtypes = [vp, i64, i64, i32, i32, i32, i32, vp, vp, vp, vp]
    L.mpreid_rerank_workspace_bytes_ex.restype = sz
    L.mpreid_rerank_workspace_bytes_ex.argtypes = [i64, i64, i32, i32, i32, i32, i32]
    L.mpreid_rerank_f32_ex.restype = i32
    L.mpreid_rerank_f32_ex.argtypes = [vp, vp, i64, i64, i32, i32, i32, f64, vp, i32, vp, i64, vp, sz, vp,
                                       C.POINTER(RerankStats), i32, i32]
    L.mpreid_rerank_debug_copy_ex.restype = i32
    L.mpreid_rerank_debug_copy_ex.argtypes = [vp, i64, i64, i32, i32, i32, i32, vp, vp, vp, vp, i32]
    L.mpreid_eval_rank_positions.restype = i32
    L.mpreid_eval_rank_positions.argtypes = [vp, i64, i32, i32, vp, vp, i32, vp, vp, vp]
    L.mpreid_rr_dist_rows.restype = i32
    L.mpreid_rr_dist_rows.argtypes = [vp, vp, i64, i32, i64, i64, vp, i64, vp, vp, i32, vp]
    L.mpreid_rr_vcap.restype = i32
    L.mpreid_rr_vcap.argtypes = [i64, i32]
    L.mpreid_rr_krecip_scratch_bytes.restype = sz
    L.mpreid_rr_krecip_scratch_bytes.argtypes = [i64]
    L.mpreid_rr_krecip.restype = i32
    L.mpreid_rr_krecip.argtypes = [vp, i64, i64, vp, vp, i32, i32, i64, i64, vp, vp, vp, vp, vp]
    L.mpreid_rr_sparse_workspace_bytes.restype = sz
    L.mpreid_rr_sparse_workspace_bytes.argtypes = [i64, i32, i64, i32]
    L.mpreid_rr_neighbours_sparse.restype = i32
    L.mpreid_rr_neighbours_sparse.argtypes = [vp, vp, i64, i32, i64, i64, i32, vp, vp, vp, vp, sz, vp]
    L.mpreid_rr_krecip_sparse.restype = i32
    L.mpreid_rr_krecip_sparse.argtypes = [vp, vp, i64, i32, vp, vp, vp, i32, i32, i64, i64, vp, vp, vp, vp, vp]
    L.mpreid_rr_pack_rows.restype = i32
    L.mpreid_rr_pack_rows.argtypes = [vp, vp, vp, i64, i32, i32, vp, vp, vp]
    L.mpreid_rr_rowptr.restype = i32
    L.mpreid_rr_rowptr.argtypes = [vp, i64, vp, vp]
    for f in (L.mpreid_rr_ell_to_csr, L.mpreid_rr_csr_to_ell):
        f.restype = i32
        f.argtypes = [vp, vp, vp, i64, i32, vp, vp, vp]
    L.mpreid_rr_qe_count.restype = i32
    L.mpreid_rr_qe_count.argtypes = [i64, vp, i32, i32, i64, i64, vp, vp, i32, vp, vp]
    L.mpreid_rr_qe_fill.restype = i32
    L.mpreid_rr_qe_fill.argtypes = [i64, vp, i32, i32, i64, i64, vp, vp, vp, i32, i32, vp, vp, vp, vp]
    L.mpreid_rr_jaccard.restype = i32
    L.mpreid_rr_jaccard.argtypes = [i64, i64, i64, i64, vp, i64, vp, vp, vp, vp, i32, f64, vp, vp, vp, vp, vp, vp, i64, vp]
    L.mpreid_rr_jaccard_hist_bytes.restype = sz
    L.mpreid_rr_jaccard_hist_bytes.argtypes = [i64]
    L.mpreid_rr_csc_chunks.restype = i32
    L.mpreid_rr_csc_chunks.argtypes = [i64, i64]
    L.mpreid_rr_csc_count.restype = i32
    L.mpreid_rr_csc_count.argtypes = [i64, i64, vp, vp, i32, i64, i64, vp, vp, vp]
    L.mpreid_rr_csc_fill.restype = i32
    L.mpreid_rr_csc_fill.argtypes = [i64, i64, vp, vp, vp, i32, i64, i64, vp, vp, vp, vp, vp, vp]
    L.mpreid_rr_jaccard_indexed.restype = i32
    L.mpreid_rr_jaccard_indexed.argtypes = [i64, i64, i64, i64, vp, i64, vp, vp, vp, vp, i32, f64, vp, vp, vp, vp, i64, vp]
    L.mpreid_vit_workspace_bytes.restype = sz
    L.mpreid_vit_workspace_bytes.argtypes = [C.POINTER(VitCfg), i32]
    L.mpreid_vit_forward.restype = i32
    L.mpreid_vit_forward.argtypes = [C.POINTER(VitCfg), C.POINTER(VitWeights), vp, i32, vp, vp, vp, sz, vp]
    L.mpreid_vit_workspace_bytes_f32.restype = sz
    L.mpreid_vit_workspace_bytes_f32.argtypes = [C.POINTER(VitCfg), i32]
    L.mpreid_vit_forward_f32.restype = i32
    L.mpreid_vit_forward_f32.argtypes = [C.POINTER(VitCfg), C.POINTER(VitWeights), vp, i32, vp, vp, vp, sz, vp]
    L.mpreid_vit_forward_f32_view.restype = i32
    L.mpreid_vit_forward_f32_view.argtypes = [C.POINTER(VitCfg), C.POINTER(VitWeights), vp, vp, C.POINTER(C.c_float),
                                              C.POINTER(C.c_float), i32, i32, vp, vp, vp, sz, vp]
    L.mpreid_vit_forward_u8.restype = i32
    L.mpreid_vit_forward_u8.argtypes = [C.POINTER(VitCfg), C.POINTER(VitWeights), vp, C.POINTER(C.c_float),
                                        C.POINTER(C.c_float), i32, vp, vp, vp, sz, vp]
    L.mpreid_vit_forward_view.restype = i32
    L.mpreid_vit_forward_view.argtypes = [C.POINTER(VitCfg), C.POINTER(VitWeights), vp, vp, C.POINTER(C.c_float),
                                          C.POINTER(C.c_float), i32, i32, vp, vp, vp, sz, vp]
    L.mpreid_tta_mean_f32.restype = i32
    L.mpreid_tta_mean_f32.argtypes = [vp, i32, i64, i32, i32, vp, vp]
    L.mpreid_resize_workspace_bytes.restype = sz
    L.mpreid_resize_workspace_bytes.argtypes = [i32, i32, i32]
    L.mpreid_resize_bilinear_u8.restype = i32
    L.mpreid_resize_bilinear_u8.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, vp, sz, vp]
    L.mpreid_rn50_workspace_bytes.restype = sz
    L.mpreid_rn50_workspace_bytes.argtypes = [C.POINTER(Rn50Cfg), i32]
    L.mpreid_rn50_forward.restype = i32
    L.mpreid_rn50_forward.argtypes = [C.POINTER(Rn50Cfg), C.POINTER(Rn50Weights), vp, vp, C.POINTER(C.c_float),
                                      C.POINTER(C.c_float), i32, vp, vp, sz, vp]
    L.mpreid_rn50_workspace_bytes_f32.restype = sz
    L.mpreid_rn50_workspace_bytes_f32.argtypes = [C.POINTER(Rn50Cfg), i32]
    L.mpreid_rn50_forward_f32.restype = i32
    L.mpreid_rn50_forward_f32.argtypes = [C.POINTER(Rn50Cfg), C.POINTER(Rn50WeightsF32), vp, i32, vp, vp, sz, vp]
    L.mpreid_rn50_workspace_bytes_split.restype = sz
    L.mpreid_rn50_workspace_bytes_split.argtypes = [C.POINTER(Rn50Cfg), i32]
    L.mpreid_rn50_forward_split.restype = i32
    L.mpreid_rn50_forward_split.argtypes = [C.POINTER(Rn50Cfg), C.POINTER(Rn50WeightsSplit), vp, i32, vp, vp, sz, vp]
    L.mpreid_rn50_forward_f32_u8.restype = i32
    L.mpreid_rn50_forward_f32_u8.argtypes = [C.POINTER(Rn50Cfg), C.POINTER(Rn50WeightsF32), vp, C.POINTER(C.c_float),
                                             C.POINTER(C.c_float), i32, vp, vp, sz, vp]
    L.mpreid_rn50_forward_split_u8.restype = i32
    L.mpreid_rn50_forward_split_u8.argtypes = [C.POINTER(Rn50Cfg), C.POINTER(Rn50WeightsSplit), vp, C.POINTER(C.c_float),
                                               C.POINTER(C.c_float), i32, vp, vp, sz, vp]
    L.mpreid_rn50_forward_f32_view.restype = i32
    L.mpreid_rn50_forward_f32_view.argtypes = [C.POINTER(Rn50Cfg), C.POINTER(Rn50WeightsF32), vp, vp, C.POINTER(C.c_float),
                                               C.POINTER(C.c_float), i32, i32, vp, vp, sz, vp]
    L.mpreid_rn50_forward_split_view.restype = i32
    L.mpreid_rn50_forward_split_view.argtypes = [C.POINTER(Rn50Cfg), C.POINTER(Rn50WeightsSplit), vp, vp, C.POINTER(C.c_float),
                                                 C.POINTER(C.c_float), i32, i32, vp, vp, sz, vp]
    L.mpreid_conv_f16_nhwc.restype = i32
    L.mpreid_conv_f16_nhwc.argtypes = [vp, i32, i32, i32, i32, vp, vp, i32, i32, i32, vp, i32, vp, vp, vp]
    L.mpreid_gemm_f16_nt.restype = i32
    L.mpreid_gemm_f16_nt.argtypes = [vp, vp, vp, i64, i64, i64, vp]
    L.mpreid_gemm_f16_nt_ex.restype = i32
    L.mpreid_gemm_f16_nt_ex.argtypes = [vp, vp, vp, vp, i64, i64, i64, i32, vp]
    L.mpreid_gemm_f16_split_nt.restype = i32
    L.mpreid_gemm_f16_split_nt.argtypes = [vp, vp, vp, vp, i64, i64, i64, f32, i32, vp]
    L.mpreid_split_pack_f32.restype = i32
    L.mpreid_split_pack_f32.argtypes = [vp, i64, i32, f32, vp, vp]
    L.mpreid_cast_f32_to_f16.restype = i32
    L.mpreid_cast_f32_to_f16.argtypes = [vp, vp, i64, vp]
    L.mpreid_profile_enable.restype = i32
    L.mpreid_profile_enable.argtypes = [i32]
    L.mpreid_profile_reset.restype = i32
    L.mpreid_profile_query.restype = i32
    L.mpreid_profile_query.argtypes = [C.POINTER(ProfileEntry), i32]
    _lib = L
    return L


def check(rc: int, what: str):
    if rc != 0:
        msg = load().mpreid_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def require_gpu():
    """The compute entry points need a HIP device; fail loudly instead of falling back."""
    import torch
    load()
    if not torch.cuda.is_available():
        raise RuntimeError("mpreid HIP path needs an MI355X (no HIP device visible); there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
