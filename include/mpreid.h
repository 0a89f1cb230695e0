/* mpreid.h — C ABI of libmpreid_hip.so: the MI355X (gfx950) implementation of MP-ReID's evaluation
 * hot path.  Plain pointers and sizes only; every pointer named "device" is HBM memory of the
 * current HIP device; every call is ordered on `stream` (a hipStream_t passed as void*).
 *
 * The reference (a pure-Python repo) has no FFI layer; the interface each entry point stands
 * behind is the Python function cited next to it.  INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add (it is the one mp-reid_amd/mpreid/_lib.py uses).
 *
 * Return value: 0 on success, otherwise a negative mpreid error or a positive hipError_t;
 * mpreid_last_error() returns a thread-local description.  Nothing falls back to the CPU.
 */
#ifndef MPREID_H
#define MPREID_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPREID_OK 0
#define MPREID_ERR_ARG (-1)
#define MPREID_ERR_WORKSPACE (-2)
#define MPREID_ERR_UNSUPPORTED (-3)
#define MPREID_ERR_NODEVICE (-4)
#define MPREID_ERR_RETRY_DENSE (-5) /* sparse re-ranking hit a data-dependent capacity: repeat with MPREID_RERANK_DENSE */

/* Environment: the library reads ONE tuning string, MPREID_TUNE="key=value,key=value" (once per process), that selects
 * between BIT-IDENTICAL forms of a stage; no other variable changes which kernel runs.  Keys (default):
 *   gemm_big (1)            0 = never the 256x256 persistent GEMM, 1 = when its grid fills the chip, 2 = whenever divisible
 *   gemm_stagger, gemm_stagger_all (0)   start delay (ticks) of the persistent workgroups
 *   gemm_walk (-1 auto)     tile order inside an XCD's group of 8 tile rows: 0 row-fastest, 1 column-fastest when the output
 *                           has <= 4 tile columns (N = 768: out-proj, FC2), 2 column-fastest always, 3 / 4 groups of 4 / 16 tile
 *                           rows; auto = column-fastest for <= 4 tile columns with operand rows >= 8 KB (the split FC2: -2.4 %);
 *                           groups of 4 rows whenever groups of 8 do not divide among the XCDs but groups of 4 do
 *   gemm_ragged (1)         XCD-owned row groups also when the groups do not divide among the XCDs (>= 24 groups: one XCD gets one
 *                           group less, a short last group leaves unused slots); 0 = round 5's rule (such shapes take the blocked walk)
 *   gemm_grid (0 = all CUs) persistent GEMM workgroups on fewer CUs (experiments: profiles/r06_cu_partition_experiment.log)
 *   dist_sym_p2 (1)         all-pairs distances of ONE tensor (q == g), fp16 modes: 0 = the 256x256 kernel's symmetric form,
 *                           1 = the two-workgroups-per-CU kernel from 16 tile rows on (N >= 3841; one-pass fp16 mode only: the 3-term
 *                           split operands stay on the first), 2 = whenever the padded size allows, 3 = also the 3-term split;
 *                           dist_sym_p2_naps / dist_sym_p2_grid (0): late start of every CU's second workgroup / grid size
 *                           (experiments); dist_sym_p2_abl: ablation variants, -DMPREID_ABLATION builds only
 *   jaccard_wave (-1 auto), jaccard_wave_rows (10240), jaccard_table (0)   form of the Jaccard stage
 *   csc_atomic (0)                       inverted index: the round-1 atomic build
 *   rerank_overlap (-1 auto)             exact query rows in line (0) / on a side stream (1)
 *   verbose (0)             occupancy messages on stderr
 * Unknown keys are reported on stderr and ignored.  Switches that change RESULTS exist only in -DMPREID_ABLATION builds. */

typedef void *mpreid_stream_t; /* hipStream_t */

int mpreid_version(void);
/* 1 when the library was compiled with -DMPREID_ABLATION (the timing-ablation switches that skip matrix instructions or
 * stores are honoured: wrong results by design), 0 for the product build.  The Python host refuses an ablation build unless
 * MPREID_ALLOW_ABLATION=1 (mpreid/_lib.py). */
int mpreid_is_ablation_build(void);
const char *mpreid_last_error(void);
/* number of visible HIP devices (0 when there is none); never throws, does not create a context */
int mpreid_device_count(void);
/* name/CU count of the current device */
int mpreid_device_info(char *name, int name_len, int *cu_count, size_t *hbm_bytes);

/* ---- distance path ----------------------------------------------------------------------- */
/* Arithmetic modes of the feat x feat^T GEMM */
#define MPREID_GEMM_F32_EXACT 0 /* v_mfma_f32_32x32x2_f32: k-ascending fmaf chain, bit-reproducible  */
#define MPREID_GEMM_F16_FAST 1  /* fp16 inputs, fp32 accumulate, one pass (|err| ~1e-4 on unit rows)  */
#define MPREID_GEMM_F16_SPLIT3 2 /* x = hi + lo (fp16 pair): hi.hi' + lo.hi' + hi.lo' on the fp16 matrix cores,
                                    fp32 accumulate; |err| <= 1e-6 on unit rows (the exact chain's own rounding
                                    level) -- meets the 1e-5 entry bound, not bit-reproducible against the oracle */

/* squared L2 norm of each row, out[n].  Order of summation is fixed (see oracle/mpreid_oracle.c). */
int mpreid_sqnorm_f32(const float *x_dev, int64_t n, int d, float *out_dev, mpreid_stream_t stream);

/* utils/metrics.py:112-114 — torch.nn.functional.normalize(feats, dim=1, p=2) with eps (1e-12). */
int mpreid_l2_normalize_f32(const float *x_dev, int64_t n, int d, float eps, float *out_dev,
                            mpreid_stream_t stream);

/* utils/metrics.py:7-13 euclidean_distance(qf, gf): out[i*ldo + j] = |q_i|^2 + |g_j|^2 - 2 q_i.g_j
 * q [nq][d], g [ng][d] fp32 row-major; out fp32, leading dimension ldo >= ng (lets a rank write its
 * gallery shard's column block of a wider matrix).  ws: mpreid_distance_workspace_bytes(). */
size_t mpreid_distance_workspace_bytes(int64_t nq, int64_t ng, int d, int mode);
int mpreid_euclidean_distance_f32(const float *q_dev, const float *g_dev, int64_t nq, int64_t ng, int d,
                                  float *out_dev, int64_t ldo, int mode, void *ws_dev, size_t ws_bytes,
                                  mpreid_stream_t stream);
/* utils/metrics.py:15-25 cosine_similarity(qf, gf): arccos(clip(q.g/(|q||g|), -1+1e-5, 1-1e-5)) */
int mpreid_cosine_similarity_f32(const float *q_dev, const float *g_dev, int64_t nq, int64_t ng, int d,
                                 float *out_dev, int64_t ldo, int mode, void *ws_dev, size_t ws_bytes,
                                 mpreid_stream_t stream);

/* ---- k-reciprocal re-ranking, utils/reranking.py:29-100 ------------------------------------ */
typedef struct {
    int64_t n;            /* nq + ng */
    int32_t k1, k2, half_k1;
    int32_t v_cap;        /* ELL row capacity of V before query expansion */
    int32_t vqe_cap;      /* row capacity after query expansion (max union size, measured) */
    int64_t v_nnz;        /* nnz(V) before query expansion */
    int64_t vqe_nnz;      /* nnz(V) after query expansion (== v_nnz when k2 == 1) */
    int64_t jaccard_pairs;/* sum over queries i, columns c in nz(V[i]) of nnz(V[:,c]) */
    int64_t krecip_r_sum; /* sum over rows of |R(i, k1)| (k-reciprocal set sizes before expansion) */
    int64_t fallback_rows;/* sparse algorithm: rows whose candidate list could not be certified (done densely) */
    int64_t cand_total;   /* sparse algorithm: neighbour candidates emitted by the fused GEMM (sum over rows) */
    int32_t algo;         /* MPREID_RERANK_DENSE / _SPARSE / _SPARSE_SPLIT3: what the call actually ran */
    /* filled when timing != 0 (hipEvents on the stream).  DENSE: gemm = N x N exact distances, topk = row maxima +
     * neighbour selection.  SPARSE: gemm = fp16 operands + sample pass + thresholds + fused candidate GEMM, topk =
     * exact refinement + fallback rows, dq = exact distance rows of the queries. */
    float ms_gemm, ms_topk, ms_krecip, ms_qe, ms_csc, ms_jaccard, ms_total, ms_dq;
} mpreid_rerank_stats;

/* Two algorithms, the same bits out (tests/test_gpu_rerank.py):
 *   DENSE   the N x N fp32 distance matrix is computed (exact fp32 MFMA), kept in HBM and scanned for the first
 *           max(k1+1, k2) neighbours of every row.  Any N, local_distmat / only_local supported.
 *   SPARSE  the N x N matrix is never materialised: a one-pass fp16 GEMM on the matrix cores emits, per row, the
 *           ~180 entries that can still be among its neighbours or be its maximum (thresholds from a 1/16 column
 *           sample, a proven error bound), those are re-evaluated with the exact fp32 chain, and the exact distance
 *           rows exist only for the queries ([nq][N], what the Jaccard blend reads).  Rows that cannot be certified
 *           fall back to the dense computation of that row.  Needs N >= 2048, max(k1+1, k2) <= 64, no local_distmat.
 *   AUTO    SPARSE when it applies, else DENSE.  A sparse call may return MPREID_ERR_RETRY_DENSE (degenerate data:
 *           too many fallback rows, a query-expansion row above 4096 entries, row norms >= 3e4): repeat it with
 *           MPREID_RERANK_DENSE and a workspace sized for DENSE. */
#define MPREID_RERANK_AUTO 0
#define MPREID_RERANK_DENSE 1
#define MPREID_RERANK_SPARSE 2
/* the sparse algorithm with the blend term's distance rows (lambda * d / max, queries x gallery) from the fp16 matrix
 * cores (3-term split, |error| <= 1e-6 on unit-norm features) instead of the exact fp32 chain: neighbour table, V, V_qe
 * and the Jaccard term are bit-identical to the other modes, |final - exact final| <= lambda * 1e-6 / max.  Never chosen
 * by AUTO. */
#define MPREID_RERANK_SPARSE_SPLIT3 3
/* Bytes of device workspace re_ranking needs for this problem (DENSE: dominated by the N x N fp32 distance matrix,
 * 4*N*N; SPARSE: by the sample distances N*N/4 bytes and the query rows 4*nq*N). */
size_t mpreid_rerank_workspace_bytes(int64_t nq, int64_t ng, int d, int k1, int k2, int has_local);      /* DENSE */
size_t mpreid_rerank_workspace_bytes_ex(int64_t nq, int64_t ng, int d, int k1, int k2, int has_local, int algo);

/* re_ranking(probFea, galFea, k1, k2, lambda_value, local_distmat=None, only_local=False)
 *   q [nq][d], g [ng][d] fp32 device; local_dev: NULL or [N][N] fp32 device (N = nq+ng);
 *   out_dev [nq][ldo] fp32 receives final_dist[:nq, nq:]  (ldo >= ng).
 * lambda is a double because the reference rounds (1 - lambda) from a Python float straight to
 * float16 and lambda to float32.  The call synchronises `stream` internally (sizes of the sparse
 * structures are read back) and returns after the result is complete.
 * stats may be NULL; with timing != 0 per-stage times are measured with hipEvents on `stream`.
 * Limits (the reference has none, utils/reranking.py:53-54; it is called with k1 = 50, k2 = 15, utils/metrics.py:127):
 *   max(k1 + 1, k2) <= 256 -- the neighbour selection sorts its winners in one 256-entry LDS network and the reciprocity
 *   masks are 256 bits per row -> MPREID_ERR_UNSUPPORTED above that;
 *   the expansion lists of one row live in LDS: 8 * min(N, (k1 + 1) * (1 + ceil(k1 / 2))) + N / 8 + 8 * (k1 + 1) + 2 KB must
 *   fit a workgroup's 160 KB, i.e. k1 <= ~190 at N >= 20 000 (k1 <= 255 for N <= 18 000) -> hipErrorInvalidValue from the
 *   launch configuration above that (a loud failure, never a wrong result);
 *   N = nq + ng < 2^31 - 64. */
/* mpreid_rerank_f32 (+ mpreid_rerank_workspace_bytes, mpreid_rerank_debug_copy) is the DENSE algorithm: it never returns
 * the data-dependent MPREID_ERR_RETRY_DENSE.  mpreid_rerank_f32_ex takes the algorithm (AUTO / SPARSE: faster, may ask for
 * the dense retry with a workspace from mpreid_rerank_workspace_bytes_ex(..., MPREID_RERANK_DENSE)). */
int mpreid_rerank_f32(const float *q_dev, const float *g_dev, int64_t nq, int64_t ng, int d, int k1, int k2,
                      double lambda_value, const float *local_dev, int only_local, float *out_dev,
                      int64_t ldo, void *ws_dev, size_t ws_bytes, mpreid_stream_t stream,
                      mpreid_rerank_stats *stats, int timing);

int mpreid_rerank_f32_ex(const float *q_dev, const float *g_dev, int64_t nq, int64_t ng, int d, int k1, int k2,
                         double lambda_value, const float *local_dev, int only_local, float *out_dev,
                         int64_t ldo, void *ws_dev, size_t ws_bytes, mpreid_stream_t stream,
                         mpreid_rerank_stats *stats, int timing, int algo);

/* debug/inspection taps used by the parity tests: copies of intermediate results of the LAST
 * mpreid_rerank_f32 call that used workspace ws_dev (valid until the workspace is reused).
 *   rank_out [N][k1+1] int32 (initial_rank[:, :k1+1]); v_cnt/vqe_cnt [N] int32 nnz per row. */
int mpreid_rerank_debug_copy(const void *ws_dev, int64_t nq, int64_t ng, int d, int k1, int k2, int has_local,
                             int32_t *rank_out_host, int32_t *v_cnt_host, int32_t *vqe_cnt_host,
                             mpreid_stream_t stream);   /* layout of the DENSE algorithm (mpreid_rerank_f32) */
int mpreid_rerank_debug_copy_ex(const void *ws_dev, int64_t nq, int64_t ng, int d, int k1, int k2, int has_local,
                                int32_t *rank_out_host, int32_t *v_cnt_host, int32_t *vqe_cnt_host,
                                mpreid_stream_t stream, int algo);

/* ---- eval_func ranking, utils/metrics.py:28-88 ---------------------------------------------------------
 * For every query: the 0-based positions, in the ascending (distance, gallery index) order of its row, of the
 * gallery items whose pid equals the query's pid (what the reference reads off np.argsort + matches).
 * pos_out [nq][rcap] int32 ascending, padded with -1; cnt_out [nq] = number of relevant items, or -1 when a
 * query has more than min(rcap, 8192) of them (LIMIT: the sorted relevant keys of a query live in LDS -- the launch sizes
 * its dynamic LDS for rcap rounded up to a power of two, 64 ... 8192 entries; the caller ranks such a row on the host --
 * utils/metrics.py:eval_func_device does -- Market-1501 / MSMT17 queries have at most a few hundred relevant gallery
 * images).  CMC / AP are finished
 * on the host from the positions (float64, numpy's summation order): mp-reid_amd/utils/metrics.py. */
int mpreid_eval_rank_positions(const float *dist_dev, int64_t ld, int nq, int ng, const int64_t *q_pids_dev,
                               const int64_t *g_pids_dev, int rcap, int32_t *pos_out_dev, int32_t *cnt_out_dev,
                               mpreid_stream_t stream);

/* ---- row-sharded re-ranking: the same kernels, phase by phase over a row range ------------
 * Rows [r_lo, r_lo+rows) of the N x N problem belong to the calling rank; between the phases the caller
 * all-gathers (RCCL) the rank table, the sparse V rows and the sparse V_qe rows.  mpreid/distributed.py
 * (re_ranking_sharded) is the driver; results do not depend on the number of ranks. */
/* phase 1: D rows + row maxima (+ first kr neighbours when rank_local != NULL); norms_all = mpreid_sqnorm_f32 */
int mpreid_rr_dist_rows(const float *feat_all_dev, const float *norms_all_dev, int64_t n, int d, int64_t r_lo,
                        int64_t rows, float *d_local_dev, int64_t ld, float *rowmax_local_dev,
                        int32_t *rank_local_dev, int kr, mpreid_stream_t stream);
/* ELL row capacity of V before query expansion: min(N, (k1+1)*(1+half_k1)) */
int mpreid_rr_vcap(int64_t n, int k1);
/* phase 2: V rows (ELL, row stride mpreid_rr_vcap) of the local rows from the global rank table [N][kr] */
/* scratch_dev: mpreid_rr_krecip_scratch_bytes(n) bytes (reciprocity bits of all n rows of the table) */
size_t mpreid_rr_krecip_scratch_bytes(int64_t n);
int mpreid_rr_krecip(const float *d_local_dev, int64_t ld, int64_t n, const float *rowmax_local_dev,
                     const int32_t *rank_all_dev, int k1, int kr, int64_t r_lo, int64_t rows, int32_t *vcnt_dev,
                     int32_t *vidx_dev, uint16_t *vval_dev, void *scratch_dev, mpreid_stream_t stream);
/* SPARSE forms of phases 1 and 2 (no [rows][n] distance block; n >= 2048, kr <= 64): phase 1 returns the first kr
 * neighbours, the row maxima and the exact distances of the neighbours of the local rows; MPREID_ERR_RETRY_DENSE
 * (degenerate data) means "use mpreid_rr_dist_rows + mpreid_rr_krecip for this row block" -- every rank may decide
 * for itself, the bits are the same.  The Jaccard phase is mpreid_rr_jaccard as before. */
size_t mpreid_rr_sparse_workspace_bytes(int64_t n, int d, int64_t rows, int kr);
int mpreid_rr_neighbours_sparse(const float *feat_all_dev, const float *norms_all_dev, int64_t n, int d, int64_t r_lo,
                                int64_t rows, int kr, int32_t *rank_local_dev, float *rowmax_local_dev,
                                float *rankd_local_dev, void *ws_dev, size_t ws_bytes, mpreid_stream_t stream);
int mpreid_rr_krecip_sparse(const float *feat_all_dev, const float *norms_all_dev, int64_t n, int d,
                            const float *rowmax_local_dev, const int32_t *rank_all_dev, const float *rankd_local_dev,
                            int k1, int kr, int64_t r_lo, int64_t rows, int32_t *vcnt_dev, int32_t *vidx_dev,
                            uint16_t *vval_dev, void *scratch_dev, mpreid_stream_t stream);
/* re-stride ELL rows (for the all-gather: common width = global max count) */
int mpreid_rr_pack_rows(const int32_t *cnt_dev, const int32_t *idx_dev, const uint16_t *val_dev, int64_t rows,
                        int src_stride, int dst_stride, int32_t *idx_out_dev, uint16_t *val_out_dev,
                        mpreid_stream_t stream);
/* CSR transport of sparse rows (what the all-gathers of V and V_qe move: nnz entries of 6 bytes instead of rows x the
 * global maximum row length): rowptr [rows + 1] = exclusive prefix sums of cnt; ell_to_csr packs the first cnt[r] entries of
 * every ELL row back to back; csr_to_ell is the inverse (entries past cnt[r] of the ELL rows are left untouched). */
int mpreid_rr_rowptr(const int32_t *cnt_dev, int64_t rows, long long *rowptr_dev, mpreid_stream_t stream);
int mpreid_rr_ell_to_csr(const long long *rowptr_dev, const int32_t *idx_dev, const uint16_t *val_dev, int64_t rows, int stride,
                         int32_t *idx_out_dev, uint16_t *val_out_dev, mpreid_stream_t stream);
int mpreid_rr_csr_to_ell(const long long *rowptr_dev, const int32_t *idx_dev, const uint16_t *val_dev, int64_t rows, int stride,
                         int32_t *idx_out_dev, uint16_t *val_out_dev, mpreid_stream_t stream);
/* phase 3: local query expansion of the local rows from the global V (row stride vstride) */
int mpreid_rr_qe_count(int64_t n, const int32_t *rank_all_dev, int kr, int k2, int64_t r_lo, int64_t rows,
                       const int32_t *vcnt_all_dev, const int32_t *vidx_all_dev, int vstride,
                       int32_t *ucnt_local_dev, mpreid_stream_t stream);
int mpreid_rr_qe_fill(int64_t n, const int32_t *rank_all_dev, int kr, int k2, int64_t r_lo, int64_t rows,
                      const int32_t *vcnt_all_dev, const int32_t *vidx_all_dev, const uint16_t *vval_all_dev,
                      int vstride, int qcap, int32_t *qcnt_local_dev, int32_t *qidx_local_dev,
                      uint16_t *qval_local_dev, mpreid_stream_t stream);
/* phase 4: inverted index of the global V_qe + Jaccard / blend for query rows [q_lo, q_lo+qrows);
 * out [qrows][ldo] = final_dist[q_lo : q_lo+qrows, nq:].  Scratch: ccnt [N+1] u32, cptr [N+1] i64,
 * crow / cval [sum of qcnt_all]. */
/* chist_dev: NULL (atomic build of the inverted index) or mpreid_rr_jaccard_hist_bytes(n) bytes: block histograms of the
 * atomics-free build, which also lets the Jaccard kernel gather exact column sub-ranges per row chunk */
size_t mpreid_rr_jaccard_hist_bytes(int64_t n);
int mpreid_rr_jaccard(int64_t n, int64_t nq, int64_t q_lo, int64_t qrows, const float *d_q_dev, int64_t ld,
                      const float *rowmax_q_dev, const int32_t *qcnt_all_dev, const int32_t *qidx_all_dev,
                      const uint16_t *qval_all_dev, int qstride, double lambda_value, uint32_t *ccnt_dev,
                      long long *cptr_dev, int32_t *crow_dev, uint16_t *cval_dev, uint32_t *chist_dev, float *out_dev,
                      int64_t ldo, mpreid_stream_t stream);
/* phase 4 with the inverted-index BUILD sharded by column range over the ranks (utils/reranking.py:80-82 is the build,
 * :84-93 the Jaccard loop): mpreid_rr_jaccard builds the whole index on every rank; here rank r
 *   1. mpreid_rr_csc_count: counts the indexed entries (rows [nq, n)) of its columns [c_lo, c_hi) -> ccnt[c_lo .. c_hi);
 *      chist (mpreid_rr_jaccard_hist_bytes(n) bytes) keeps per-block offsets for step 2          -> ALL-GATHER ccnt [n] u32
 *   2. mpreid_rr_csc_fill: cptr [n + 1] = exclusive scan of the gathered counts (ccnt_all is consumed), the packed entries of
 *      its columns at their GLOBAL positions cpk[cptr[c_lo] .. cptr[c_hi]) (cpk: cptr[n] u32 words) and the rows
 *      [c_lo, c_hi) of the chunk-boundary table hb [n][mpreid_rr_csc_chunks(n, nq) + 1] u32
 *                                                       -> ALL-GATHER the cpk pieces (contiguous, rank order) and the hb rows
 *   3. mpreid_rr_jaccard_indexed: Jaccard + blend of the rank's queries over the assembled index.
 * Needs n * qstride < 2^32 and at least one indexed row (0 <= nq < n): all three entry points check both and return
 * MPREID_ERR_ARG otherwise (mpreid_rr_csc_chunks returns the chunk count, > 0, or that negative code).
 * Bit-identical to mpreid_rr_jaccard for any number of column shards. */
int mpreid_rr_csc_chunks(int64_t n, int64_t nq);
int mpreid_rr_csc_count(int64_t n, int64_t nq, const int32_t *qcnt_all_dev, const int32_t *qidx_all_dev, int qstride,
                        int64_t c_lo, int64_t c_hi, uint32_t *chist_dev, uint32_t *ccnt_dev, mpreid_stream_t stream);
int mpreid_rr_csc_fill(int64_t n, int64_t nq, const int32_t *qcnt_all_dev, const int32_t *qidx_all_dev,
                       const uint16_t *qval_all_dev, int qstride, int64_t c_lo, int64_t c_hi, uint32_t *ccnt_all_dev,
                       uint32_t *chist_dev, long long *cptr_dev, uint32_t *cpk_dev, uint32_t *hb_dev, mpreid_stream_t stream);
int mpreid_rr_jaccard_indexed(int64_t n, int64_t nq, int64_t q_lo, int64_t qrows, const float *d_q_dev, int64_t ld,
                              const float *rowmax_q_dev, const int32_t *qcnt_all_dev, const int32_t *qidx_all_dev,
                              const uint16_t *qval_all_dev, int qstride, double lambda_value, const long long *cptr_dev,
                              const uint32_t *cpk_dev, const uint32_t *hb_dev, float *out_dev, int64_t ldo,
                              mpreid_stream_t stream);

/* ---- CLIP ViT-B/16 image encoder, model/clip/model.py:415-479 + model/make_model.py:81-115 -- */
typedef struct {
    int32_t img_h, img_w;   /* INPUT.SIZE_TEST */
    int32_t patch, stride;  /* 16, MODEL.STRIDE_SIZE[0] */
    int32_t h_res, w_res;   /* patch grid; tokens L = h_res*w_res + 1 */
    int32_t width, layers, heads, out_dim; /* 768, 12, 12, 512 */
    int32_t neck_after;     /* TEST.NECK_FEAT == 'after': apply the eval BatchNorm necks */
    int32_t cls_only_last;  /* 1: last block computes only the CLS row (the only row the output uses) */
    int32_t precision;      /* MPREID_VIT_F16 or MPREID_VIT_SPLIT (the all-fp32 mode is mpreid_vit_forward_f32) */
} mpreid_vit_cfg;

/* Arithmetic of the encoder's linear layers and attention products (model/clip/model.py runs them in fp32,
 * processor/processor.py:187-198 has no autocast):
 *   F16    fp16 operands, fp32 accumulate: the fast mode; relative feature error ~4e-4 (operand rounding of 24 GEMMs),
 *          which moves mAP by 1e-4..3e-4 on hard data -- does NOT meet the 1e-4 mAP bound.
 *   SPLIT  every operand is an fp16 PAIR x = hi + lo (22 significant bits) and every product runs as
 *          hi.hi' + lo.hi' + hi.lo' with fp32 accumulation on the same fp16 matrix cores: fp32-grade results (relative
 *          feature error ~1e-6, |dmAP| <= 1e-4) at 3x the matrix work.  The parity mode that is also the measured mode.
 * In SPLIT mode every *_w pointer of the weight structs is an fp16 pair matrix [out][2*in] = [hi(in) | lo(in)] of
 * W * 2^e (e per matrix, so that the largest |entry| is in [2^9, 2^10)), and the matching *_s field is 2^-e
 * (mpreid_split_pack_f32 produces the layout).
 * Range of the fp16 modes (both): ACTIVATIONS are split / rounded unscaled, so every GEMM input -- the normalised pixels, the
 * LayerNorm outputs, q / k / v, the softmax-weighted values, the QuickGELU outputs -- must stay below 65 504 in magnitude.
 * Above that a `hi` half is +-inf, the products turn into NaN and the NaN reaches the image's feature row through the
 * residual stream (it is NOT clamped or hidden): R1_mAP_eval.compute() tests the features and raises RuntimeError
 * ("non-finite") instead of ranking them; use the all-fp32 mode for such a checkpoint / input range.  Trained CLIP towers
 * stay 1-2 orders of magnitude below the bound where it applies: the +-2000 "massive activation" channels live in the
 * fp32 residual stream, which is never an fp16 operand; LayerNorm outputs are at most sqrt(width) * max|gamma| + max|beta|.
 * Small values: an element below 2^-3 in magnitude keeps an ABSOLUTE error of 2^-25 (its `lo` half is an fp16 subnormal),
 * which is the level of fp32 rounding noise relative to O(1) activations; tests/test_gpu_vit.py pins both edges
 * (test_split_mode_on_clip_like_outliers: x30-100 LayerNorm gammas, +-1900 residual channels, +-60 FC1 pre-activations within
 * 2e-5 of the float64 graph and within a few times plain fp32's own error; test_split_mode_input_scale_edges). */
#define MPREID_VIT_F16 0
#define MPREID_VIT_SPLIT 1
/* (precision 2 was MPREID_VIT_SPLIT_LNFOLD in round 3: the LayerNorms folded into the linear layers behind them.  Removed in
 * round 4 -- measured 0.5 % SLOWER than MPREID_VIT_SPLIT, and tools/ws_poison_check.py found its 128 x 128-kernel form not
 * reproducible run to run at small batches; mpreid_vit_forward returns MPREID_ERR_UNSUPPORTED for it.  The in_proj_c / fc_c
 * fields below are kept for layout compatibility and are not read.) */

typedef struct { /* device pointers; *_w are fp16 [out][in] row-major (torch Linear layout) */
    const void *in_proj_w;   /* [3*width][width] fp16 */
    const float *in_proj_b;  /* [3*width] */
    const void *out_proj_w;  /* [width][width] fp16 */
    const float *out_proj_b;
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    const void *fc_w;        /* [4*width][width] fp16 */
    const float *fc_b;
    const void *proj_w;      /* [width][4*width] fp16 */
    const float *proj_b;
    float in_proj_s, out_proj_s, fc_s, proj_s;   /* SPLIT mode: 2^-e of the matching weight matrix (ignored in F16 mode) */
    const float *in_proj_c, *fc_c;               /* not read (see MPREID_VIT_SPLIT above) */
} mpreid_vit_layer;

typedef struct { /* device pointers */
    const void *conv_w;         /* [width][3*patch*patch] fp16, inner order (c, kh, kw) */
    const float *class_emb;     /* [width] */
    const float *pos_emb;       /* [L][width] */
    const float *ln_pre_g, *ln_pre_b, *ln_post_g, *ln_post_b;
    const float *proj;          /* [width][out_dim] fp32 */
    const float *bn_scale, *bn_shift;           /* [width]   eval BN folded: y = x*scale + shift (or NULL) */
    const float *bn_proj_scale, *bn_proj_shift; /* [out_dim] */
    const mpreid_vit_layer *layers;             /* HOST array of cfg.layers entries */
    float conv_s;                               /* SPLIT mode: 2^-e of conv_w */
} mpreid_vit_weights;

size_t mpreid_vit_workspace_bytes(const mpreid_vit_cfg *cfg, int batch);
/* img_dev [B][3][img_h][img_w] fp32 (already mean/std normalised); cv_emb_dev NULL or [B][width]
 * (SIE_COE * cv_embed[idx], model/make_model.py:89-96); out_dev [B][width+out_dim] fp32. */
int mpreid_vit_forward(const mpreid_vit_cfg *cfg, const mpreid_vit_weights *w, const float *img_dev, int batch,
                       const float *cv_emb_dev, float *out_dev, void *ws_dev, size_t ws_bytes,
                       mpreid_stream_t stream);

/* All-fp32 mode of the same encoder (parity / debugging; SURVEY.md section 7 hard part 5): activations and weights
 * fp32, linear layers on the exact fp32 matrix instruction, attention on fp32 vector FMAs -- relative feature error
 * ~1e-6 against the fp32 CPU path instead of ~4e-4, at ~1/8 of the throughput.  The structs are the ones above,
 * but conv_w and every layer's *_w point to FP32 [out][in] matrices.  Head dimension 64 only. */
size_t mpreid_vit_workspace_bytes_f32(const mpreid_vit_cfg *cfg, int batch);
int mpreid_vit_forward_f32(const mpreid_vit_cfg *cfg, const mpreid_vit_weights *w_f32, const float *img_dev, int batch,
                           const float *cv_emb_dev, float *out_dev, void *ws_dev, size_t ws_bytes,
                           mpreid_stream_t stream);
/* the all-fp32 mode on uint8 input and / or one test-time-augmentation view (MPREID_VIEW_* below): exactly one of
 * img_f32_dev ([B][3][H][W], already normalised) and img_hwc_u8_dev ([B][H][W][3] + pixel_mean3 / pixel_std3 host arrays) is
 * non-NULL; ToTensor + Normalize and the view transform happen inside the patch gather, in the reference's arithmetic */
int mpreid_vit_forward_f32_view(const mpreid_vit_cfg *cfg, const mpreid_vit_weights *w_f32, const float *img_f32_dev,
                                const uint8_t *img_hwc_u8_dev, const float *pixel_mean3, const float *pixel_std3, int view,
                                int batch, const float *cv_emb_dev, float *out_dev, void *ws_dev, size_t ws_bytes,
                                mpreid_stream_t stream);

/* same, from uint8 images [B][img_h][img_w][3] (HWC, after Resize): ToTensor (x/255) and Normalize
 * ((x - pixel_mean)/pixel_std, host arrays of 3 floats: INPUT.PIXEL_MEAN / PIXEL_STD) of the reference's
 * val_transforms (datasets/make_dataloader.py:57-61) are fused into the patch gather; 4x fewer input bytes. */
int mpreid_vit_forward_u8(const mpreid_vit_cfg *cfg, const mpreid_vit_weights *w, const uint8_t *img_hwc_dev,
                          const float *pixel_mean3, const float *pixel_std3, int batch, const float *cv_emb_dev,
                          float *out_dev, void *ws_dev, size_t ws_bytes, mpreid_stream_t stream);

/* Test-time-augmentation views of the reference's Uni-Prompt evaluation ("option A",
 * processor/processor_uniprompt_stage2.py:605-633), applied while the patches are gathered instead of
 * materialising a transformed [B,3,H,W] tensor per view. */
#define MPREID_VIEW_ORIGINAL 0
#define MPREID_VIEW_FLIP 1        /* torch.flip(img, [3]) */
#define MPREID_VIEW_PSEUDO_IR 2   /* img.mean(dim=1, keepdim=True).repeat(1, 3, 1, 1) -- in the HOST arithmetic of torch.mean,
                                   * ((c0 + c1) + c2) / 3 with a correctly rounded division: the reference's CPU path, bit for
                                   * bit.  torch's device kernel multiplies the sum by a rounded 1/3 (<= 1 ulp apart per pixel):
                                   * features of a device-materialised view agree to ~1e-6, not to the bit
                                   * (tests/test_gpu_preprocess.py::test_pseudo_ir_view_against_a_device_materialised_view) */
#define MPREID_VIEW_PSEUDO_RGB 3  /* img[:, 0:1].repeat(1, 3, 1, 1) */
/* mpreid_vit_forward / mpreid_vit_forward_u8 on one view: exactly one of img_f32_dev ([B][3][H][W], already
 * normalised) and img_hwc_u8_dev ([B][H][W][3] + pixel_mean3 / pixel_std3 host arrays) is non-NULL. */
int mpreid_vit_forward_view(const mpreid_vit_cfg *cfg, const mpreid_vit_weights *w, const float *img_f32_dev,
                            const uint8_t *img_hwc_u8_dev, const float *pixel_mean3, const float *pixel_std3, int view,
                            int batch, const float *cv_emb_dev, float *out_dev, void *ws_dev, size_t ws_bytes,
                            mpreid_stream_t stream);
/* processor/processor_uniprompt_stage2.py:636-640: out[rows][dim] = mean over the views of
 * feats[n_views][rows][dim] (sequential sum in view order, true division), then F.normalize(p=2, dim=1,
 * eps=1e-12) when normalize != 0. */
int mpreid_tta_mean_f32(const float *feats_dev, int n_views, int64_t rows, int dim, int normalize, float *out_dev,
                        mpreid_stream_t stream);

/* T.Resize(cfg.INPUT.SIZE_TEST) of val_transforms (datasets/make_dataloader.py:57-58) for a ragged batch of decoded
 * uint8 RGB images: bit-exact with PIL.Image.resize((out_w, out_h), BILINEAR) (Pillow 8-bit two-pass resample,
 * which is what torchvision's Resize calls for PIL inputs).  Image b is src_dev[offsets_dev[b] ...] as [h][w][3]
 * with (h, w) = hw_dev[2b], hw_dev[2b+1]; max_in_h >= every h.  dst_dev is [batch][out_h][out_w][3], the input
 * layout of mpreid_vit_forward_u8. */
size_t mpreid_resize_workspace_bytes(int batch, int max_in_h, int out_w);
int mpreid_resize_bilinear_u8(const uint8_t *src_dev, const int64_t *offsets_dev, const int32_t *hw_dev, int batch,
                              int max_in_h, int out_h, int out_w, uint8_t *dst_dev, void *ws_dev, size_t ws_bytes,
                              mpreid_stream_t stream);

/* ---- CLIP "RN50" image encoder (MODEL.NAME == 'RN50': model/clip/model.py:10-148 ModifiedResNet, Bottleneck,
 * AttentionPool2d) + the RN50 eval branch of build_transformer.forward (model/make_model.py:82-86, 102-115):
 * out = cat(avg_pool2d(x4), attnpool(x4)[0]) (after the eval BatchNorm necks for NECK_FEAT == 'after').
 * NHWC fp16 activations, every conv an implicit MFMA GEMM with its BatchNorm folded in (see
 * mpreid_conv_f16_nhwc), residual add + ReLU in the GEMM epilogue. ---- */
typedef struct { /* one folded convolution; device pointers */
    const void *w;        /* fp16 [cout_pad][taps*cin], k order (kh, kw, c) */
    const float *bias;    /* fp32 [cout_pad] */
    int32_t cin, cout, cout_pad, taps;  /* cin % 64 == 0, cout % 8 == 0, cout_pad % 128 == 0, taps 1 or 9 */
} mpreid_rn50_conv;

typedef struct { /* Bottleneck (model/clip/model.py:10-53) */
    mpreid_rn50_conv conv1, conv2, conv3, down;  /* down.w == NULL when the block has no downsample branch */
    int32_t stride;                              /* AvgPool2d(stride) after conv2 and in front of down: 1 or 2 */
} mpreid_rn50_block;

typedef struct {
    int32_t img_h, img_w;   /* multiples of 32 */
    int32_t width;          /* 64: stem 32/32/64 channels, layers 64/128/256/512 planes */
    int32_t n_blocks;       /* sum of the layers tuple (3+4+6+3 = 16) */
    int32_t heads, out_dim; /* attention pool: embed dim = 32*width, heads, output_dim (1024) */
} mpreid_rn50_cfg;

typedef struct { /* device pointers unless noted */
    const float *stem1_w;   /* conv1+bn1 folded, fp32 [width/2][3][3][3] = [cout][c][kh][kw] */
    const float *stem1_b;   /* fp32 [width/2] */
    mpreid_rn50_conv stem2, stem3;          /* cin = cout = 64 storage channels (32 real ones in stem2) */
    const mpreid_rn50_block *blocks;        /* HOST array of n_blocks entries */
    const float *pos_emb;                   /* attnpool.positional_embedding fp32 [S+1][E] */
    const void *kt_w;                       /* fp16 [E][E] = k_proj.weight TRANSPOSED (k_proj.bias cancels in the softmax) */
    const void *v_w; const float *v_b;      /* fp16 [E][E] = v_proj.weight, fp32 [E] */
    const void *q_w; const float *q_b;      /* fp16 [E][E], fp32 [E] */
    const void *c_w; const float *c_b;      /* fp16 [out_pad128][E], fp32 [out_pad128] */
    const float *bn_scale, *bn_shift;       /* eval BN necks folded, fp32 [E + out_dim], or NULL */
} mpreid_rn50_weights;

size_t mpreid_rn50_workspace_bytes(const mpreid_rn50_cfg *cfg, int batch);
/* exactly one of img_f32_dev ([B][3][H][W] fp32, val_transforms applied) and img_hwc_u8_dev ([B][H][W][3] uint8 +
 * pixel_mean3 / pixel_std3 host arrays: ToTensor + Normalize fused into the first convolution) is non-NULL;
 * out_dev [B][32*width + out_dim] fp32. */
int mpreid_rn50_forward(const mpreid_rn50_cfg *cfg, const mpreid_rn50_weights *w, const float *img_f32_dev,
                        const uint8_t *img_hwc_u8_dev, const float *pixel_mean3, const float *pixel_std3, int batch,
                        float *out_dev, void *ws_dev, size_t ws_bytes, mpreid_stream_t stream);

/* All-fp32 mode of the RN50 tower (MODEL.NAME 'RN50' + MODEL.ENCODER_PRECISION 'fp32'; the reference runs the tower in
 * fp32): fp32 NHWC activations, every convolution a GEMM on the exact fp32 matrix instruction (3x3: im2col, k order
 * (kh, kw, c)), BatchNorm folded on the host, fp32 attention pool -- relative feature error ~1e-6 against the fp32 CPU path
 * instead of the fp16 tower's 2.6e-3, at ~1/10 of its throughput.  Weight matrices are fp32 [cout][taps*cin] with the REAL
 * channel counts (no padding); input fp32 NCHW only. */
typedef struct { const float *w; const float *bias; int32_t cin, cout, taps; } mpreid_rn50_conv_f32;
typedef struct { mpreid_rn50_conv_f32 conv1, conv2, conv3, down; int32_t stride; } mpreid_rn50_block_f32;  /* down.w NULL: none */
typedef struct {
    const float *stem1_w, *stem1_b;              /* conv1+bn1 folded, [width/2][3][3][3], [width/2] */
    mpreid_rn50_conv_f32 stem2, stem3;
    const mpreid_rn50_block_f32 *blocks;         /* HOST array of n_blocks entries */
    const float *pos_emb;                        /* [S+1][E] */
    const float *q_w, *q_b, *k_w, *k_b, *v_w, *v_b;   /* [E][E], [E] */
    const float *c_w, *c_b;                      /* [out_dim][E], [out_dim] */
    const float *bn_scale, *bn_shift;            /* eval BN necks folded, [E + out_dim], or NULL */
} mpreid_rn50_weights_f32;
size_t mpreid_rn50_workspace_bytes_f32(const mpreid_rn50_cfg *cfg, int batch);
int mpreid_rn50_forward_f32(const mpreid_rn50_cfg *cfg, const mpreid_rn50_weights_f32 *w, const float *img_f32_dev, int batch,
                            float *out_dev, void *ws_dev, size_t ws_bytes, mpreid_stream_t stream);
/* the same tower fed with uint8 [batch][H][W][3] images (after Resize): ToTensor + Normalize of
 * datasets/make_dataloader.py:57-61, (x / 255 - mean[c]) / std[c] with correctly rounded divisions, inside the stem's first
 * convolution -- same bits as the host-transformed fp32 tensor, a quarter of the input bytes, no intermediate tensor */
int mpreid_rn50_forward_f32_u8(const mpreid_rn50_cfg *cfg, const mpreid_rn50_weights_f32 *w, const uint8_t *img_u8_hwc_dev,
                               const float *mean3, const float *std3, int batch, float *out_dev, void *ws_dev, size_t ws_bytes,
                               mpreid_stream_t stream);

/* SPLIT-precision mode of the RN50 tower (MODEL.NAME 'RN50' + MODEL.ENCODER_PRECISION 'split', the default): fp32 NHWC
 * activations; the convolutions of layer1-4 and the attention pool's k / v projections run on the fp16 matrix cores over
 * fp16 PAIRS hi + lo, three products per multiply-add with fp32 accumulation (the GE_S_* GEMMs of the ViT's split mode):
 * fp32-grade features at several times the all-fp32 mode's throughput.  The stem's first convolution (K = 27), the one-query attention and the 1-row
 * projections q / c stay on the exact fp32 path (`f32`: stem1_*, pos_emb, q_*, c_*, bn_* are read; its stem2 / stem3 /
 * blocks / k_* / v_* are not).  A split convolution: w = fp16 pair matrix [npad][2 * kseg] = [hi(kseg) | lo(kseg)] of the
 * BatchNorm-folded weights [cout][taps * cin] (k order (kh, kw, c)) times 2^e, zero-padded to kseg = taps * cin rounded up to
 * 64 columns and npad = cout rounded up to 128 rows (mpreid_split_pack_f32); bias fp32 [npad] (zero past cout);
 * oscale = 2^-e.  cin % 4 == 0.  Range: as for the ViT's split mode (activations below 65 504).
 * taps == 9 (the 3x3 convolutions run as IMPLICIT GEMMs over the nine shifted views of the pair activations, no im2col
 * matrix): kseg = cin rounded up to 64 (the padded channel count of ONE tap) and w = fp16 slabs
 * [npad][tap = kh*3+kw][kseg / 64][hi(64) | lo(64)] of W * 2^e (channels >= cin and rows >= cout zero). */
typedef struct { const void *w; const float *bias; int32_t cin, cout, taps, kseg, npad; float oscale; } mpreid_rn50_conv_split;
typedef struct { mpreid_rn50_conv_split conv1, conv2, conv3, down; int32_t stride; } mpreid_rn50_block_split;   /* down.w NULL: none */
typedef struct {
    mpreid_rn50_weights_f32 f32;
    const mpreid_rn50_block_split *blocks;       /* HOST array of n_blocks entries */
    mpreid_rn50_conv_split k, v;                 /* attention pool k_proj / v_proj as 1x1 "convolutions" over the tokens */
    mpreid_rn50_conv_split stem2, stem3;         /* the stem's second and third convolution (f32.stem2 / stem3 are not read) */
} mpreid_rn50_weights_split;
size_t mpreid_rn50_workspace_bytes_split(const mpreid_rn50_cfg *cfg, int batch);
int mpreid_rn50_forward_split(const mpreid_rn50_cfg *cfg, const mpreid_rn50_weights_split *w, const float *img_f32_dev, int batch,
                              float *out_dev, void *ws_dev, size_t ws_bytes, mpreid_stream_t stream);
int mpreid_rn50_forward_split_u8(const mpreid_rn50_cfg *cfg, const mpreid_rn50_weights_split *w, const uint8_t *img_u8_hwc_dev,
                                 const float *mean3, const float *std3, int batch, float *out_dev, void *ws_dev, size_t ws_bytes,
                                 mpreid_stream_t stream);   /* uint8 input, as mpreid_rn50_forward_f32_u8 */
/* one test-time-augmentation view (MPREID_VIEW_*) of fp32 [B][3][H][W] or uint8 [B][H][W][3] input (exactly one pointer
 * non-NULL), the view transform applied to the normalised pixels inside the stem's first convolution: the same bits as the
 * materialised view tensor of processor/processor_uniprompt_stage2.py:605-633 */
int mpreid_rn50_forward_split_view(const mpreid_rn50_cfg *cfg, const mpreid_rn50_weights_split *w, const float *img_f32_dev,
                                   const uint8_t *img_u8_hwc_dev, const float *mean3, const float *std3, int view, int batch,
                                   float *out_dev, void *ws_dev, size_t ws_bytes, mpreid_stream_t stream);
int mpreid_rn50_forward_f32_view(const mpreid_rn50_cfg *cfg, const mpreid_rn50_weights_f32 *w, const float *img_f32_dev,
                                 const uint8_t *img_u8_hwc_dev, const float *mean3, const float *std3, int view, int batch,
                                 float *out_dev, void *ws_dev, size_t ws_bytes, mpreid_stream_t stream);

/* One convolution layer of the RN50 tower as the encoder runs it (unit tests, micro-benchmarks):
 * NHWC fp16 in [batch][h][w][cin] (cin % 64 == 0), stride 1, taps = 1 (1x1) or 9 (3x3, pad 1); weights fp16
 * [cout_pad][taps*cin] (k order: tap = kh*3+kw, then channel; cout_pad % 128 == 0, rows >= cout zero) with the
 * BatchNorm of model/clip/model.py:17-24 folded in, bias fp32 [cout_pad]; optional fp16 identity [M][cout] added
 * before the ReLU (Bottleneck.forward, model/clip/model.py:39-53); out fp16 [batch*h*w][cout]; zero_page_dev =
 * 128 bytes of zeros. */
int mpreid_conv_f16_nhwc(const void *act_dev, int batch, int h, int w, int cin, const void *wgt_dev, const float *bias_dev,
                         int cout, int cout_pad, int taps, const void *identity_dev, int relu, void *out_dev,
                         const void *zero_page_dev, mpreid_stream_t stream);

/* fp16 GEMM used by the encoder, exposed for the roofline bench and unit tests:
 * C[M][N] (fp32) = A[M][K] (fp16) x B[N][K]^T (fp16).  M, N multiples of 128... see DESIGN.md. */
int mpreid_gemm_f16_nt(const void *a_dev, const void *b_dev, float *c_dev, int64_t m, int64_t n, int64_t k,
                       mpreid_stream_t stream);
/* same GEMM with a fused epilogue: 0 = fp32 out, 1 = +bias -> fp16, 2 = out(fp32) += acc + bias,
 * 3 = QuickGELU(acc + bias) -> fp16, 7 = relu(acc + bias) -> fp16, 8 = out(fp16) = relu(acc + bias + out), one
 * rounding (the RN50 1x1 convolutions).  Used by the unit tests and tools/gemm_bench.py. */
int mpreid_gemm_f16_nt_ex(const void *a_dev, const void *b_dev, void *out_dev, const float *bias_dev, int64_t m,
                          int64_t n, int64_t k, int epilogue, mpreid_stream_t stream);
/* The linear layer of the encoder's SPLIT precision mode, exposed for unit tests and tools/gemm_bench.py:
 * a2 [M][2*kseg], b2 [N][2*kseg] fp16 pairs [hi | lo]; acc = sum over k of hi.hi' + lo.hi' + hi.lo' (fp32);
 * epilogue 10: out fp32 [M][N] = acc*oscale + bias; 11: out fp32 [M][N] += acc*oscale + bias;
 * 12: out fp16 pair [M][2N] = hi | lo of QuickGELU(acc*oscale + bias). */
int mpreid_gemm_f16_split_nt(const void *a2_dev, const void *b2_dev, void *out_dev, const float *bias_dev, int64_t m,
                             int64_t n, int64_t kseg, float oscale, int epilogue, mpreid_stream_t stream);
/* x [rows][cols] fp32 -> y [rows][2*cols] fp16 pair: hi = fp16(x*scale), lo = fp16(x*scale - hi); scale a power of two */
int mpreid_split_pack_f32(const float *x_dev, int64_t rows, int cols, float scale, void *y_dev, mpreid_stream_t stream);
/* fp32 -> fp16 (RNE) conversion of a flat array */
int mpreid_cast_f32_to_f16(const float *x_dev, void *y_dev, int64_t n, mpreid_stream_t stream);

/* ---- measurement hooks (bench.py roofline leg) --------------------------------------------- */
/* kernel classes of the encoder that are not fp16 GEMMs (mpreid_profile_entry.epilogue); flops_total then holds
 * ALGORITHMIC BYTES for the HBM-bound ones (layer norm: 4 B read + output bytes per element; attention: q, k, v read +
 * output written) */
#define MPREID_PROF_LAYERNORM 100
#define MPREID_PROF_ATTENTION 101
#define MPREID_PROF_EVALRANK 102   /* eval_rank_kernel: m = nq, n = ng; work = 4*nq*ng bytes (the matrix read once) */
typedef struct {
    int32_t epilogue;        /* GemmEpi id of the fp16 GEMM class, or MPREID_PROF_* */
    int32_t n, k;            /* GEMM N and K (split epilogues: K = 2 * kseg halfs per operand row; 3 * kseg is executed) */
    int64_t m;               /* GEMM M (padded rows): classes are keyed by (epilogue, M, N, K) */
    int64_t launches;        /* launches recorded while profiling was enabled */
    double total_ms;         /* sum of per-launch durations (hipEvents on the launch stream) */
    double flops_total;      /* sum over the recorded launches of 2*M*N*K */
} mpreid_profile_entry;
/* While enabled, every fp16 GEMM launch is bracketed by two hipEvents on its stream. */
int mpreid_profile_enable(int on);
int mpreid_profile_reset(void);
/* Synchronises the recorded events; entries sorted by total time, returns the class count. */
int mpreid_profile_query(mpreid_profile_entry *out, int cap);

#ifdef __cplusplus
}
#endif
#endif /* MPREID_H */
