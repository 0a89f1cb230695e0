// rerank.hip — k-reciprocal re-ranking on gfx950 (reference: utils/reranking.py:29-100).
//
// Data layout in HBM (all inside the caller's workspace, see make_layout()):
//   feat  [N][d] fp32          cat(probFea, galFea)                                  (:36)
//   D     [N][ld] fp32         all-pairs squared distance, exact fp32 MFMA GEMM      (:36-41)
//                              ld = N rounded up to 64 floats; D is bit-symmetric
//   MT    [N][ld] fp32         only with local_distmat: (D + local)^T                (:43-46)
//   rowmax[N]                  max over column i of original_dist == row max of MT   (:46)
//   rank  [N][KR] int32        first KR = max(k1+1, k2) entries of argsort(O[i,:])   (:48)
//                              O[i][j] = MT[i][j] / rowmax[i] is recomputed where needed
//   V     ELL  idx[N][vcap] int32 ascending, val[N][vcap] fp16 bits, cnt[N]           (:51-71)
//   Vqe   ELL  same, row stride = measured max union size                             (:73-78)
//   CSC of Vqe: cptr[N+1] int64, crow[nnz] int32, cval[nnz] fp16 bits                  (:80-82)
//   out   [nq][ldo] fp32       final_dist[:nq, nq:]                                   (:84-100)
//
// Every rounding point of SURVEY.md §8a row a7 is taken through include/mpreid_numerics.h, the
// same code the CPU oracle runs, so the result is bit-identical to oracle/mpreid_oracle.c.
// All kernels here are HBM/L2-bound integer+fp16 work (no MFMA); the GEMM is in distance.hip.
#include <algorithm>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <vector>

#include "common.h"
#include "gemm_f16.h"

int mpreid_distance_launch(const float *q, const float *g, int64_t nq, int64_t ng, int d, const float *qn,
                           const float *gn, float *out, int64_t ldo, int epi, hipStream_t stream,
                           const unsigned *m_count = nullptr);
// gemm_f16.hip: 3-term fp16-split distances (|error| <= 1e-6 on unit-norm rows)
int mpreid_distance_f16_split3(const float *q, const float *g, int64_t nq, int64_t ng, int d, const float *qn,
                               const float *gn, float *out, int64_t ldo, int epi, void *ws, size_t ws_bytes,
                               hipStream_t stream);
size_t mpreid_distance_split3_ws_bytes(int64_t nq, int64_t ng, int d);

// ---------------------------------------------------------------------------------------------
// Native binary16 on the device.  include/mpreid_numerics.h spells numpy's float16 arithmetic out in integer code
// (fp32 operation, then a branchy RNE conversion: ~50 instructions), which is what the oracle compiles.  On gfx950
// v_cvt_f16_f32 / v_cvt_f32_f16 / v_add_f16 are IEEE round-to-nearest-even with subnormals (the f16 denormal
// mode is always on), and "add in fp32, round to fp16" equals one correctly rounded fp16 add (double rounding is
// innocuous when the wide format has >= 2p+2 = 24 significand bits).  Same bits, one instruction: the Jaccard
// accumulation was ALU-bound on the integer form.  tests/test_gpu_rerank.py compares every output bit with the
// oracle, so a device on which this did not hold would fail there.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint16_t h_from_f32(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
__device__ __forceinline__ float h_to_f32(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
// numpy's float16 sub / mul / div, literally: the operation in fp32, one RNE conversion to fp16
__device__ __forceinline__ uint16_t h_sub_native(uint16_t a, uint16_t b) { return h_from_f32(h_to_f32(a) - h_to_f32(b)); }
__device__ __forceinline__ uint16_t h_mul_native(uint16_t a, uint16_t b) { return h_from_f32(h_to_f32(a) * h_to_f32(b)); }
__device__ __forceinline__ uint16_t h_div_native(uint16_t a, uint16_t b) {
    return h_from_f32(__fdiv_rn(h_to_f32(a), h_to_f32(b)));
}
__device__ __forceinline__ uint16_t h_add_native(uint16_t a, uint16_t b) {
    return __builtin_bit_cast(uint16_t, (_Float16)(__builtin_bit_cast(_Float16, a) + __builtin_bit_cast(_Float16, b)));
}

// ---------------------------------------------------------------------------------------------
// workspace layout
// ---------------------------------------------------------------------------------------------
// How the Jaccard stage cuts the row space of the blocked inverted index (256 row blocks of rows_per_block rows): a
// workgroup accumulates one chunk of blocks_per_chunk blocks (rch rows of fp16 accumulators in LDS).  A function of N
// only, so that the workspace layouts can reserve the chunk-boundary table ([N][nchunks + 1] u32) the build writes.
struct JaccardPlan {
    int rpb, bpc, nchunks, rch;
    bool wave_form;   // one 64-thread workgroup per (query, chunk) instead of 256 threads
};
static JaccardPlan jaccard_plan(int64_t N) {   // N = number of INDEXED rows (the gallery rows)
    constexpr int B = 256;   // = CSC_B
    static const int jwave = mpreid_tune("jaccard_wave", -1);
    static const int jrows = mpreid_tune("jaccard_wave_rows", 10240);
    JaccardPlan p;
    p.rpb = (int)((N + B - 1) / B);
    // 256-thread form: at most ~24 K rows per chunk (48 KB of accumulators: two workgroups per CU beside the tables)
    p.bpc = std::max(1, std::min(B, 24576 / p.rpb));
    p.nchunks = (B + p.bpc - 1) / p.bpc;
    // more than one chunk: ONE wave per (query, chunk of ~10 K rows) -- no barriers at all (the LDS operations of a wave
    // execute in order, which is all the fp16 accumulation order needs), LDS = the accumulators only (jaccard_wave_kernel;
    // measured at N = 100 000 / MSMT17 shape: 4 K rows 15.6 / 10.2 ms, 6 K 14.6 / 8.6, 8 K 14.1 / 8.9, 10 K 13.4 / 8.6,
    // 12 K 15.6 / 9.2, 16 K 19.2 / 14.4)
    p.wave_form = jwave < 0 ? p.nchunks > 1 : jwave > 0;
    if (p.wave_form) {
        p.bpc = std::max(1, std::min(B, jrows / p.rpb));
        p.nchunks = (B + p.bpc - 1) / p.bpc;
    }
    p.rch = (int)align_up((size_t)std::min<int64_t>(N, (int64_t)p.bpc * p.rpb), 8);
    return p;
}
// bytes of the block histograms [256][N] + the chunk-boundary table [N][nchunks + 1]
// (sized for the worst case over the number of indexed rows: nchunks <= 256)
static size_t csc_hist_bytes(int64_t N) { return ((size_t)N * 256 + (size_t)N * 257) * 4 + ((size_t)(N + 1023) / 1024 + 2) * 8; }

struct RerankLayout {
    int64_t N, ld;
    int K, KR, h, vcap;
    int64_t qcap_bound;
    size_t feat, norms, D, MT, rowmax, rank, rbits, vcnt, vidx, vval, ucnt, qcnt, qidx, qval, ccnt, chist, cptr, crow, cval,
        counters, total;
};

static RerankLayout make_layout(int64_t nq, int64_t ng, int d, int k1, int k2, int has_local) {
    RerankLayout L{};
    L.N = nq + ng;
    L.ld = (int64_t)align_up((size_t)L.N, 64);
    // numpy slicing clamps the neighbour lists when N is smaller than k1+1 / k1/2+1 / k2 (utils/reranking.py:53-54,
    // 60-62,76); np.mean then averages over min(k2, N) rows
    L.K = (int)std::min<int64_t>(k1 + 1, L.N);
    L.KR = std::max(L.K, (int)std::min<int64_t>(k2, L.N));
    L.h = (int)std::min<int64_t>(mpreid_half_k1(k1), L.N);
    int64_t cap = (int64_t)L.K * (1 + L.h);
    L.vcap = (int)std::min<int64_t>(cap, L.N);
    L.qcap_bound = std::min<int64_t>(L.N, (int64_t)std::max(k2, 1) * L.vcap);
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += align_up(bytes, 256);
        return o;
    };
    const size_t N = (size_t)L.N;
    L.feat = take(N * (size_t)d * 4);
    L.norms = take(N * 4);
    L.D = take(N * (size_t)L.ld * 4);
    L.MT = has_local ? take(N * (size_t)L.ld * 4) : L.D;
    L.rowmax = take(N * 4);
    L.rank = take(N * (size_t)L.KR * 4);
    L.rbits = take(N * (size_t)8 * 4 * 2);  // RB_WORDS words per row, two masks
    L.vcnt = take(N * 4);
    L.vidx = take(N * (size_t)L.vcap * 4);
    L.vval = take(N * (size_t)L.vcap * 2);
    L.ucnt = take(N * 4);
    L.qcnt = take(N * 4);
    L.qidx = take(N * (size_t)L.qcap_bound * 4);
    L.qval = take(N * (size_t)L.qcap_bound * 2);
    L.ccnt = take((N + 1) * 4);
    L.chist = take(csc_hist_bytes((int64_t)N));   // CSC_B block histograms + chunk bounds
    L.cptr = take((N + 1) * 8);
    L.crow = take(N * (size_t)L.qcap_bound * 4);
    L.cval = take(N * (size_t)L.qcap_bound * 2);
    L.counters = take(64);
    L.total = off;
    return L;
}


// ---------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------
// Inclusive scan over the 64 lanes with DPP moves (row_shr 1 / 2 / 4 / 8 inside the rows of 16 lanes, then row_bcast:15
// into rows 1 and 3 and row_bcast:31 into rows 2 and 3): six VALU operations.  The __shfl_up formulation compiles to
// ds_bpermute_b32 -- an LDS crossbar round trip per step, ~850 cycles per scan measured with in-kernel stamps: the ten
// scans of extract_bits_sorted were a third of a k-reciprocal row at N = 20 000, the 49 of the query expansion most of it
// at N = 100 000.
__device__ __forceinline__ int wave_incl_scan(int v) {
    int x = v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);    // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);    // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);    // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);    // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    return x;
}
__device__ __forceinline__ int wave_excl_scan(int v, int lane, int &total) {
    (void)lane;
    const int x = wave_incl_scan(v);
    total = __builtin_amdgcn_readlane(x, 63);
    return x - v;
}

// orderable key of a float: ascending unsigned order == ascending float order (-0 folded into +0)
__device__ __forceinline__ uint32_t fkey(float f) {
    f = f + 0.0f;
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// numpy pairwise sum of a[0..n) (LDS or global), evaluated redundantly by every 8-lane group of
// the calling wave; all lanes return the same value.  Mirrors orc_pairwise_sum_f32.
__device__ float wave_pairwise_sum(const float *a, int n, int lane) {
    const int j = lane & 7;
    // explicit post-order traversal of the recursion  pw(o,n) = pw(o,n2) + pw(o+n2,n-n2)
    int st_off[24], st_n[24];
    float st_val[24];
    unsigned char st_state[24];
    int sp = 0;
    st_off[0] = 0; st_n[0] = n; st_state[0] = 0;
    float ret = 0.0f;
    while (sp >= 0) {
        const int o = st_off[sp], m = st_n[sp];
        if (st_state[sp] == 0) {
            if (m <= 128) {
                float res;
                if (m < 8) {
                    res = 0.0f;
                    for (int i = 0; i < m; ++i) res = res + a[o + i];
                } else {
                    float r = a[o + j];
                    const int lim = m - (m % 8);
                    int i;
                    for (i = 8; i < lim; i += 8) r = r + a[o + i + j];
                    r = r + __shfl_xor(r, 1, 64);
                    r = r + __shfl_xor(r, 2, 64);
                    r = r + __shfl_xor(r, 4, 64);
                    res = r;
                    for (i = lim; i < m; ++i) res = res + a[o + i];
                }
                ret = res;
                --sp;
            } else {
                int n2 = m / 2;
                n2 -= n2 % 8;
                st_state[sp] = 1;
                ++sp;
                st_off[sp] = o; st_n[sp] = n2; st_state[sp] = 0;
            }
        } else if (st_state[sp] == 1) {
            int n2 = m / 2;
            n2 -= n2 % 8;
            st_val[sp] = ret;
            st_state[sp] = 2;
            ++sp;
            st_off[sp] = o + n2; st_n[sp] = m - n2; st_state[sp] = 0;
        } else {
            ret = st_val[sp] + ret;
            --sp;
        }
    }
    return ret;
}

// out[0] = sum of x[0..n) (one workgroup; statistics only)
__global__ __launch_bounds__(1024) void sum_i32_kernel(const int *__restrict__ x, int64_t n, unsigned long long *__restrict__ out) {
    __shared__ unsigned long long part[16];
    unsigned long long s = 0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) s += (unsigned long long)(unsigned)x[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < 16; ++w) t += part[w];
        out[0] = t;
    }
}

// ---------------------------------------------------------------------------------------------
// (D + local)^T  /  local^T  (only when local_distmat is given; utils/reranking.py:33-34,43-46)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void make_mt_kernel(const float *__restrict__ D, int64_t ldD,
                                                      const float *__restrict__ local, int64_t N,
                                                      float *__restrict__ MT, int64_t ld) {
    __shared__ float tile[32][33];
    const int64_t bx = (int64_t)blockIdx.x * 32, by = (int64_t)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int64_t i = by + r, j = bx + tx;
        float v = 0.0f;
        if (i < N && j < N) {
            v = local[i * N + j];
            if (D) v = D[i * ldD + j] + v; // original_dist + local_distmat
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int64_t i = bx + r, j = by + tx; // MT[i][j] = orig[j][i]
        if (i < N && j < N) MT[i * ld + j] = tile[tx][r];
    }
}

// ---------------------------------------------------------------------------------------------
// row max + top-KR selection by (O value, index) ascending; one 256-thread workgroup per row.
// HBM/L2-bound: the row is read 5 times (max, 3 radix histograms, collect); 4*N bytes algorithmic.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int block_excl_scan_256(int v, int tid, int *s_wave, int &total) {
    const int lane = tid & 63, wave = tid >> 6;
    int wtot;
    const int ex = wave_excl_scan(v, lane, wtot);
    __syncthreads();
    if (lane == 0) s_wave[wave] = wtot;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int t = s_wave[w];
        if (w < wave) base += t;
        tot += t;
    }
    total = tot;
    return base + ex;
}

constexpr int TOPK_CAND = 1024;
__global__ __launch_bounds__(256) void rowmax_topk_kernel(const float *__restrict__ MT, int64_t ld, int64_t N, int KR,
                                                          float *__restrict__ rowmax, int *__restrict__ rank,
                                                          const unsigned *__restrict__ row_count) {
    if (row_count && blockIdx.x >= *row_count) return;   // rows past a device-side count (fallback rows) are skipped
    __shared__ unsigned hist[2048];
    __shared__ unsigned long long sel[256];
    __shared__ unsigned long long cand[TOPK_CAND];   // (key, index) of the entries that share the first 22 key bits with
    __shared__ unsigned s_ncand, s_nsel;             // the KR-th smallest, gathered during the last radix pass
    __shared__ float s_red[4];
    __shared__ int s_wave[4];
    __shared__ unsigned s_bin, s_below, s_cnt;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t i = blockIdx.x;
    const float *row = MT + i * ld;
    const int n = (int)N;

    // pass 0: row max (== max over column i of original_dist)
    float mx = -3.402823466e+38f;
    for (int j = tid; j < n; j += 256) mx = fmaxf(mx, row[j]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    if (lane == 0) s_red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    if (tid == 0) rowmax[i] = mx;

    if (tid == 0) {
        s_ncand = 0;
        s_nsel = 0;
    }
    for (int b = tid; b < 256; b += 256) sel[b] = ~0ull;
    // radix select of the KR-th smallest key, 11 + 11 + 10 bits.  The last pass also gathers what the collection
    // needs -- entries below the 22-bit prefix class go straight to the selection, the class itself (normally a few
    // dozen entries) into cand[] -- so that the row is read four times instead of five
    unsigned prefix = 0;
    int kk = KR; // 1-based rank still to find inside the current prefix class
    int bits_done = 0;
    unsigned cnt_eq = 0;
#pragma unroll 1
    for (int pass = 0; pass < 3; ++pass) {
        const int width = (pass == 2) ? 10 : 11;
        const int shift = 32 - bits_done - width;
        const unsigned mask = (1u << width) - 1u;
        for (int b = tid; b < 2048; b += 256) hist[b] = 0;
        __syncthreads();
        for (int j = tid; j < n; j += 256) {
            const unsigned key = fkey(__fdiv_rn(row[j], mx));
            if (bits_done == 0 || (key >> (32 - bits_done)) == prefix) atomicAdd(&hist[(key >> shift) & mask], 1u);
            if (pass == 2) {
                const unsigned top = key >> 10;
                if (top < prefix) {
                    const unsigned p = atomicAdd(&s_nsel, 1u);
                    if (p < 256u) sel[p] = ((unsigned long long)key << 32) | (unsigned)j;
                } else if (top == prefix) {
                    const unsigned p = atomicAdd(&s_ncand, 1u);
                    if (p < (unsigned)TOPK_CAND) cand[p] = ((unsigned long long)key << 32) | (unsigned)j;
                }
            }
        }
        __syncthreads();
        // each thread owns 8 consecutive bins
        unsigned loc[8];
        int sum = 0;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            loc[b] = hist[tid * 8 + b];
            sum += (int)loc[b];
        }
        int total;
        const int ex = block_excl_scan_256(sum, tid, s_wave, total);
        if (kk > ex && kk <= ex + sum) {
            int run = ex;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                if (kk > run && kk <= run + (int)loc[b]) {
                    s_bin = (unsigned)(tid * 8 + b);
                    s_below = (unsigned)run;
                    s_cnt = loc[b];
                }
                run += (int)loc[b];
            }
        }
        __syncthreads();
        prefix = (prefix << width) | s_bin;
        kk -= (int)s_below;
        cnt_eq = s_cnt;
        bits_done += width;
        __syncthreads();
    }
    const unsigned T = prefix; // exact key of the KR-th smallest; kk of the cnt_eq equal keys are needed

    // collect (key, index) of the KR selected entries
    const bool from_lds = ((int)cnt_eq == kk) && s_ncand <= (unsigned)TOPK_CAND;
    __syncthreads();
    if (from_lds) {
        // no tie straddles the cut and the prefix class fitted cand[]: everything below the class is in sel[]
        // already, the class contributes its keys <= T; order fixed later by the sort
        const unsigned nc = s_ncand;
        for (unsigned c = tid; c < nc; c += 256) {
            const unsigned long long e = cand[c];
            if ((unsigned)(e >> 32) <= T) {
                const unsigned p = atomicAdd(&s_nsel, 1u);
                if (p < 256u) sel[p] = e;
            }
        }
    } else {
    if (tid == 0) s_cnt = 0;
    for (int b = tid; b < 256; b += 256) sel[b] = ~0ull;
    __syncthreads();
    if ((int)cnt_eq == kk) {
        // no tie straddles the cut: take every key <= T, order fixed later by the sort
        for (int j = tid; j < n; j += 256) {
            const unsigned key = fkey(__fdiv_rn(row[j], mx));
            if (key <= T) {
                const unsigned p = atomicAdd(&s_cnt, 1u);
                if (p < 256u) sel[p] = ((unsigned long long)key << 32) | (unsigned)j;
            }
        }
    } else {
        // ties at the cut: the kk smallest INDICES among the equal keys are taken (ordered scans)
        int run_acc = 0, run_eq = 0;
        for (int j0 = 0; j0 < n; j0 += 256) {
            const int j = j0 + tid;
            unsigned key = 0xffffffffu;
            bool lt = false, eq = false;
            if (j < n) {
                key = fkey(__fdiv_rn(row[j], mx));
                lt = key < T;
                eq = key == T;
            }
            int eq_tot, acc_tot;
            const int eq_rank = run_eq + block_excl_scan_256(eq ? 1 : 0, tid, s_wave, eq_tot);
            const bool accept = lt || (eq && eq_rank < kk);
            const int p = run_acc + block_excl_scan_256(accept ? 1 : 0, tid, s_wave, acc_tot);
            if (accept && p < 256) sel[p] = ((unsigned long long)key << 32) | (unsigned)j;
            run_acc += acc_tot;
            run_eq += eq_tot;
        }
    }
    }
    __syncthreads();
    // bitonic sort of 256 64-bit keys (padding = ~0) ascending => (value, index) order
    for (int size = 2; size <= 256; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const int partner = tid ^ stride;
            if (partner > tid) {
                const unsigned long long a = sel[tid], b = sel[partner];
                const bool up = ((tid & size) == 0);
                if ((a > b) == up) {
                    sel[tid] = b;
                    sel[partner] = a;
                }
            }
            __syncthreads();
        }
    }
    if (tid < KR) rank[i * KR + tid] = (int)(unsigned)(sel[tid] & 0xffffffffull);
}

// Register-resident variant for N <= 256 * CHUNK: the row is read from HBM exactly once (4*N bytes, the
// algorithmic minimum); every thread keeps its CHUNK orderable keys in VGPRs and the row maximum, the three
// radix-histogram passes and the collection run from registers (+ LDS atomics for the histograms).
template <int CHUNK>
__global__ __launch_bounds__(256) void rowmax_topk_reg_kernel(const float *__restrict__ MT, int64_t ld, int64_t N, int KR,
                                                              float *__restrict__ rowmax, int *__restrict__ rank,
                                                              const unsigned *__restrict__ row_count) {
    if (row_count && blockIdx.x >= *row_count) return;
    __shared__ unsigned long long sel[256];
    __shared__ float s_red[4];
    __shared__ int s_wave[4];
    __shared__ int s_part[8];
    __shared__ unsigned s_cnt;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t i = blockIdx.x;
    const float *row = MT + i * ld;
    const int n = (int)N;

    // register slot s = 4*c4 + e holds element j = c4*1024 + tid*4 + e: 16-byte loads (rows start on 256-byte
    // boundaries: ld is a multiple of 64 floats), every wave instruction moves 1 KB
    static_assert(CHUNK % 4 == 0, "CHUNK must be a multiple of 4");
    float v[CHUNK];
    float mx = -3.402823466e+38f;
#pragma unroll
    for (int c4 = 0; c4 < CHUNK / 4; ++c4) {
        const int j0 = c4 * 1024 + tid * 4;
        float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j0 + 3 < n) {
            f = *reinterpret_cast<const float4 *>(row + j0);
        } else {
            if (j0 + 0 < n) f.x = row[j0 + 0];
            if (j0 + 1 < n) f.y = row[j0 + 1];
            if (j0 + 2 < n) f.z = row[j0 + 2];
        }
        v[c4 * 4 + 0] = f.x; v[c4 * 4 + 1] = f.y; v[c4 * 4 + 2] = f.z; v[c4 * 4 + 3] = f.w;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (j0 + e < n) mx = fmaxf(mx, v[c4 * 4 + e]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    if (lane == 0) s_red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    if (tid == 0) rowmax[i] = mx;
    unsigned key[CHUNK];
#pragma unroll
    for (int c = 0; c < CHUNK; ++c) {
        const int j = (c >> 2) * 1024 + tid * 4 + (c & 3);
        key[c] = (j < n) ? fkey(__fdiv_rn(v[c], mx)) : 0xffffffffu;
    }

    // K-th smallest key by bisection on the key value: no atomics (normalised distances share their sign and
    // exponent bits, so radix histograms pile every LDS atomic onto a handful of bins).  Each iteration counts
    // the register-resident keys <= mid and reduces the count over the workgroup; 32 uniform iterations.
    // workgroup sum with ONE barrier per call: partial sums go to alternating slot sets, so a wave that runs
    // ahead writes the other set (it cannot get two calls ahead: it would have to pass the next barrier)
    int bs_phase = 0;
    auto block_sum = [&](int x) -> int {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
        int *slot = s_part + (bs_phase & 1) * 4;
        ++bs_phase;
        if (lane == 0) slot[wave] = x;
        __syncthreads();
        return (slot[0] + slot[1]) + (slot[2] + slot[3]);
    };
    unsigned lo = 0u, hi = 0xfffffffeu; // padded slots hold 0xffffffff and are never counted
#pragma unroll 1
    for (int it = 0; it < 32; ++it) {
        const unsigned mid = lo + ((hi - lo) >> 1);
        int c_le = 0;
#pragma unroll
        for (int c = 0; c < CHUNK; ++c) c_le += (key[c] <= mid) ? 1 : 0;
        if (block_sum(c_le) >= KR) hi = mid; else lo = mid + 1u;
    }
    const unsigned T = lo; // smallest key with count(key <= T) >= KR == the KR-th smallest key
    int c_lt = 0, c_eq = 0;
#pragma unroll
    for (int c = 0; c < CHUNK; ++c) {
        c_lt += (key[c] < T) ? 1 : 0;
        c_eq += (key[c] == T) ? 1 : 0;
    }
    const int cnt_lt = block_sum(c_lt);
    const unsigned cnt_eq = (unsigned)block_sum(c_eq);
    const int kk = KR - cnt_lt; // how many of the cnt_eq keys equal to T are taken
    if (tid == 0) s_cnt = 0;
    sel[tid] = ~0ull;
    __syncthreads();
    if ((int)cnt_eq == kk) {
#pragma unroll
        for (int c = 0; c < CHUNK; ++c) {
            const int j = (c >> 2) * 1024 + tid * 4 + (c & 3);
            if (j < n && key[c] <= T) {
                const unsigned p = atomicAdd(&s_cnt, 1u);
                if (p < 256u) sel[p] = ((unsigned long long)key[c] << 32) | (unsigned)j;
            }
        }
    } else {
        // ties at the cut (rare): the kk smallest INDICES among the equal keys, by ordered scans over
        // index = c*256 + tid.  Keys are recomputed from memory here: indexing the register array with a
        // loop variable would push the whole array into scratch for the common path as well.
        int run_acc = 0, run_eq = 0;
#pragma unroll 1
        for (int c = 0; c < CHUNK; ++c) {
            const int j = c * 256 + tid;
            const bool in = j < n;
            const unsigned kc = in ? fkey(__fdiv_rn(row[j], mx)) : 0xffffffffu;
            const bool lt = in && kc < T, eq = in && kc == T;
            int eq_tot, acc_tot;
            const int eq_rank = run_eq + block_excl_scan_256(eq ? 1 : 0, tid, s_wave, eq_tot);
            const bool accept = lt || (eq && eq_rank < kk);
            const int p = run_acc + block_excl_scan_256(accept ? 1 : 0, tid, s_wave, acc_tot);
            if (accept && p < 256) sel[p] = ((unsigned long long)kc << 32) | (unsigned)j;
            run_acc += acc_tot;
            run_eq += eq_tot;
        }
    }
    __syncthreads();
    if (KR <= 64) {
        // exactly KR <= 64 entries were selected: one wave sorts them with a shuffle-only bitonic network
        if (wave == 0) {
            unsigned long long x = sel[lane];
#pragma unroll
            for (int size = 2; size <= 64; size <<= 1)
#pragma unroll
                for (int stride = size >> 1; stride > 0; stride >>= 1) {
                    const unsigned lo32 = __shfl_xor((unsigned)x, stride, 64);
                    const unsigned hi32 = __shfl_xor((unsigned)(x >> 32), stride, 64);
                    const unsigned long long y = ((unsigned long long)hi32 << 32) | lo32;
                    const bool up = ((lane & size) == 0);
                    const bool lower = ((lane & stride) == 0);
                    // the lower lane of a pair keeps the smaller key in an ascending block, the larger otherwise
                    const bool keep_min = (lower == up);
                    x = keep_min ? (x < y ? x : y) : (x > y ? x : y);
                }
            if (lane < KR) rank[i * KR + lane] = (int)(unsigned)(x & 0xffffffffull);
        }
        return;
    }
    for (int size = 2; size <= 256; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const int partner = tid ^ stride;
            if (partner > tid) {
                const unsigned long long a = sel[tid], b = sel[partner];
                const bool up = ((tid & size) == 0);
                if ((a > b) == up) {
                    sel[tid] = b;
                    sel[partner] = a;
                }
            }
            __syncthreads();
        }
    }
    if (tid < KR) rank[i * KR + tid] = (int)(unsigned)(sel[tid] & 0xffffffffull);
}

static int launch_rowmax_topk(const float *MT, int64_t ld, int64_t N, int KR, int64_t rows, float *rowmax, int *rank,
                              hipStream_t stream, const unsigned *row_count = nullptr) {
    // measured on MI355X: the register-resident kernel wins for N <= 8192 (0.25 vs 0.34 ms at N = 8000); with 80+
    // keys per thread it drops to 2 waves/SIMD and loses to the radix kernel that re-reads rows from L2
    // (3.3 vs 1.9 ms at N = 20 000)
    const int64_t per = (N + 255) / 256;
    if (per <= 32)
        hipLaunchKernelGGL(rowmax_topk_reg_kernel<32>, dim3((unsigned)rows), dim3(256), 0, stream, MT, ld, N, KR, rowmax, rank,
                           row_count);
    else
        hipLaunchKernelGGL(rowmax_topk_kernel, dim3((unsigned)rows), dim3(256), 0, stream, MT, ld, N, KR, rowmax, rank,
                           row_count);
    LAUNCH_CHECK();
    return MPREID_OK;
}

// ---------------------------------------------------------------------------------------------
// k-reciprocal sets + 2/3-overlap expansion + exp weights -> V row (ELL).  One wave per row.
// utils/reranking.py:51-71
// ---------------------------------------------------------------------------------------------
// wpre (optional): wpre[w] = number of set bits below word w, written for the words of every non-empty group of 64 (the
// only ones a set bit can live in): slot(c) = wpre[c >> 5] + popc(mask[c >> 5] & below(c))
__device__ __forceinline__ int extract_bits_sorted(const unsigned *mask, int nw, int *list, int lane, int *wpre = nullptr) {
    int base = 0;
    for (int w0 = 0; w0 < nw; w0 += 64) {
        const int w = w0 + lane;
        unsigned word = (w < nw) ? mask[w] : 0u;
        if (__ballot(word != 0u) == 0ull) continue;
        int tot;
        int pos = base + wave_excl_scan(__popc(word), lane, tot);
        if (wpre && w < nw) wpre[w] = pos;
        while (word) {
            const int b = __ffs((int)word) - 1;
            word &= word - 1u;
            list[pos++] = w * 32 + b;
        }
        base += tot;
    }
    return base;
}

// exact squared distance of rows i and j of `feat`, the oracle's arithmetic (oracle/mpreid_oracle.c: dot_block +
// orc_euclid): a k-ascending fmaf chain from 0, then fmaf(-2, dot, |f_i|^2 + |f_j|^2) -- the same bits as one entry
// of the exact distance GEMM.  qrow: row i (LDS or global), grow: row j in global memory.
__device__ __forceinline__ float exact_dist_chain(const float *qrow, const float *__restrict__ grow, int d, float ni, float nj) {
    float acc = 0.0f;
    int k = 0;
    if ((reinterpret_cast<uintptr_t>(grow) & 15) == 0 && d >= 32) {
        // one 128-byte line of row j per batch (8 x 16 B issued back to back: the lanes of a wave walk 64 different
        // rows, so a batch touches 64 lines once), the next batch in flight while this one feeds the chain
        float4 nx[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) nx[u] = *reinterpret_cast<const float4 *>(grow + 4 * u);
        const int nb = d >> 5;
        for (int b = 0; b < nb; ++b) {
            float4 cur[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) cur[u] = nx[u];
            if (b + 1 < nb) {
#pragma unroll
                for (int u = 0; u < 8; ++u) nx[u] = *reinterpret_cast<const float4 *>(grow + (b + 1) * 32 + 4 * u);
            }
            const float *qk = qrow + b * 32;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc = fmaf(qk[4 * u + 0], cur[u].x, acc);
                acc = fmaf(qk[4 * u + 1], cur[u].y, acc);
                acc = fmaf(qk[4 * u + 2], cur[u].z, acc);
                acc = fmaf(qk[4 * u + 3], cur[u].w, acc);
            }
        }
        k = nb << 5;
    }
    for (; k < d; ++k) acc = fmaf(qrow[k], grow[k], acc);
    return fmaf(-2.0f, acc, ni + nj);
}

// Wave-cooperative form of exact_dist_chain: lane l receives the exact distance of row i (qrow, in LDS) to ITS
// candidate row jl (jl < 0: none).  The 64 candidate rows are fetched 32 floats at a time with COALESCED loads (8
// lanes take one 128-byte line of one candidate; a per-lane walk of its own row costs 64 tag look-ups per load
// instruction and ran the texture path at ~0.4 requests per clock), laid into a wave-private LDS tile [64][36] and
// read back by the owning lane, whose fmaf chain therefore keeps the oracle's k-ascending order.  The loads of the next
// 32 floats are in flight while the current ones feed the chain.  tile: 64 * 36 floats of LDS per wave.
constexpr int WXD_STRIDE = 36;
__device__ __forceinline__ float wave_exact_dists(const float *qrow, const float *__restrict__ feat, int d, int jl, float ni,
                                                  const float *__restrict__ sqn, float *tile, int lane) {
    float acc = 0.0f;
    int k0 = 0;
    const bool fast = (d % 4 == 0) && ((reinterpret_cast<uintptr_t>(feat) & 15) == 0) && d >= 32;
    if (fast) {
        const int piece = lane & 7, sub = lane >> 3;
        const float *src[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int jc = __shfl(jl, 8 * u + sub, 64);
            src[u] = (jc >= 0) ? feat + (int64_t)jc * d + piece * 4 : nullptr;
        }
        const int nb = d >> 5;
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[u] ? *reinterpret_cast<const float4 *>(src[u]) : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int b = 0; b < nb; ++b) {
#pragma unroll
            for (int u = 0; u < 8; ++u) *reinterpret_cast<float4 *>(tile + (8 * u + sub) * WXD_STRIDE + piece * 4) = v[u];
            if (b + 1 < nb) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    v[u] = src[u] ? *reinterpret_cast<const float4 *>(src[u] + (b + 1) * 32) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const float *qk = qrow + b * 32;
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const float4 g4 = *reinterpret_cast<const float4 *>(tile + lane * WXD_STRIDE + p * 4);
                acc = fmaf(qk[4 * p + 0], g4.x, acc);
                acc = fmaf(qk[4 * p + 1], g4.y, acc);
                acc = fmaf(qk[4 * p + 2], g4.z, acc);
                acc = fmaf(qk[4 * p + 3], g4.w, acc);
            }
            __builtin_amdgcn_wave_barrier();
        }
        k0 = nb << 5;
    }
    if (jl < 0) return 0.0f;
    const float *grow = feat + (int64_t)jl * d;
    for (int k = k0; k < d; ++k) acc = fmaf(qrow[k], grow[k], acc);
    return fmaf(-2.0f, acc, ni + sqn[jl]);
}

// MT / rowmax / vcnt / vidx / vval are indexed by the LOCAL row (blockIdx.x); `rank` is the global table and
// row0 the global index of local row 0 (0 on a single GPU; the rank's first row when rows are sharded).
// SPARSE (candidate pipeline, no N x N matrix): the distances of row i to its expansion set are evaluated on the fly
// from the features (exact_dist_chain); MT is unused, feat [N][d] / norms [N] are the inputs.
// ---------------------------------------------------------------------------------------------
// Reciprocity bits of the neighbour table, once per row (they depend on the table only, not on which row's
// expansion asks): for row c and position a < K with n = rank[c][a], p = position of c in rank[n][0..K) or none;
//   bit a of rk[c]  <=>  p exists            (n is a k-reciprocal neighbour of c:        R(c, k1))
//   bit a of rh[c]  <=>  a < h and p < h     (n is a k1/2-reciprocal neighbour of c:     R(c, k1/2))
// utils/reranking.py:53-58 and :61-66 evaluate exactly these with np.where per (row, candidate) pair; done per pair
// on the GPU that was 47 x 26 x 26 scattered 4-byte loads per row and the kernel ran at the texture unit's request
// rate.  RB_WORDS 32-bit words per row and mask (K <= 256).  One wave per row, neighbour rows read 16 bytes per load.
// ---------------------------------------------------------------------------------------------
constexpr int RB_WORDS = 8;
struct __attribute__((packed, aligned(4))) int4u { int v[4]; };   // 16-byte load from a 4-byte aligned address
__global__ __launch_bounds__(256) void recip_bits_kernel(const int *__restrict__ rank, int64_t N, int K, int KR, int h,
                                                         unsigned *__restrict__ rk, unsigned *__restrict__ rh) {
    const int lane = threadIdx.x & 63;
    const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= N) return;
    for (int a0 = 0; a0 < K; a0 += 64) {
        const int a = a0 + lane;
        int p = -1;
        if (a < K) {
            const int n = rank[c * KR + a];
            const int *row = rank + (int64_t)n * KR;
            int b = 0;
            for (; b + 3 < K; b += 4) {
                const int4u q = *reinterpret_cast<const int4u *>(row + b);
#pragma unroll
                for (int e = 3; e >= 0; --e) p = (q.v[e] == (int)c) ? b + e : p;   // rows hold distinct indices
            }
            for (; b < K; ++b) p = (row[b] == (int)c) ? b : p;
        }
        const unsigned long long mk = __ballot(p >= 0), mh = __ballot(p >= 0 && p < h && a < h);
        if (lane == 0) {
            rk[c * RB_WORDS + (a0 >> 5)] = (unsigned)mk;
            rh[c * RB_WORDS + (a0 >> 5)] = (unsigned)mh;
            if (a0 + 32 < RB_WORDS * 32) {
                rk[c * RB_WORDS + (a0 >> 5) + 1] = (unsigned)(mk >> 32);
                rh[c * RB_WORDS + (a0 >> 5) + 1] = (unsigned)(mh >> 32);
            }
        }
    }
}

// One 256-thread workgroup per row.  (First version: one wave per row walking the expansion candidates one after the
// other -- ~47 iterations of two dependent global loads with 26 active lanes, 130 us per row.  Now a half-wave per
// candidate, eight candidates per iteration.)  Wave 0 keeps the ordered steps (rank-order compaction of R, bitmap
// extraction, numpy-order pairwise sum, ordered write of the V row).
constexpr int KR_HASH_BITS = 9, KR_HASH = 1 << KR_HASH_BITS;   // >= 2 x the largest K (256)
// dynamic LDS of krecip_kernel (the layout at the top of the kernel)
static size_t krecip_lds_bytes(bool sparse, int nw, int K, int vcap, int d) {
    size_t b = (size_t)nw * 4 + KR_HASH * 4 + (size_t)K * 8 + (size_t)vcap * 8;
    if (sparse) b += (size_t)K * 4 + (size_t)vcap * 2 + 16 + (size_t)((d + 3) & ~3) * 4 + 64 * WXD_STRIDE * 4;
    return b;
}
// (launch bound: four workgroups per CU = at most 128 VGPRs; unconstrained hipcc took 222 for the rarely taken exact-
// distance path and two workgroups per CU -- 0.53 -> 0.35 ms at N = 20 000 with the bound, 240 B of scratch per lane)
template <bool SPARSE>
__global__ __launch_bounds__(256, 4) void krecip_kernel(const float *__restrict__ MT, int64_t ld, int64_t N,
                                                     const float *__restrict__ rowmax, const int *__restrict__ rank,
                                                     int K, int KR, int h, int vcap, int *__restrict__ vcnt,
                                                     int *__restrict__ vidx, uint16_t *__restrict__ vval, int row0,
                                                     int *__restrict__ r_count,
                                                     const float *__restrict__ feat, const float *__restrict__ norms,
                                                     int d, const float *__restrict__ rankd,
                                                     const unsigned *__restrict__ rk, const unsigned *__restrict__ rh) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nw = (int)((N + 31) >> 5);
    // LDS (krecip_lds_bytes): the expansion bit mask over all N rows, a 512-slot hash set of R (membership tests of the
    // expansion; a second N-bit mask cost two workgroups per CU at N = 100 000), the lists
    unsigned *Emask = (unsigned *)smem;
    int *Rhash = (int *)(Emask + nw);          // [KR_HASH] open addressing, -1 = empty
    int *fwd = Rhash + KR_HASH;
    int *R = fwd + K;
    int *Elist = R + K;
    float *wbuf = (float *)(Elist + vcap);
    // SPARSE only: exact distances of the first K neighbours (from the refinement), the indices of expansion entries
    // that are not among them, row i of feat and the wave_exact_dists tile (16-byte aligned)
    float *fdist = wbuf + vcap;
    uint16_t *miss = (uint16_t *)(fdist + K);  // positions in Elist (< vcap < 65536)
    float *qrow = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(miss + vcap) + 15) & ~(uintptr_t)15);
    float *xtile = qrow + ((d + 3) & ~3);
    auto rh_slot = [](int f) -> unsigned { return ((unsigned)f * 2654435761u) >> (32 - KR_HASH_BITS); };
    __shared__ int s_nR, s_nE, s_nmiss;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = blockIdx.x;      // local row
    const int i = row0 + li;        // global row

    for (int w = tid; w < nw; w += 256) Emask[w] = 0u;
    for (int w = tid; w < KR_HASH; w += 256) Rhash[w] = -1;
    for (int a = tid; a < K; a += 256) {
        fwd[a] = rank[(int64_t)i * KR + a];
        if (SPARSE) fdist[a] = rankd[(int64_t)i * KR + a];
    }
    if (tid == 0) s_nmiss = 0;
    __syncthreads();

    // k_reciprocal_index: fwd[a] such that i is among the first K neighbours of fwd[a]; rank order kept (wave 0)
    if (wave == 0) {
        int nR = 0;
        for (int a0 = 0; a0 < K; a0 += 64) {
            const int a = a0 + lane;
            bool f = false;
            int c = -1;
            if (a < K) {
                c = fwd[a];
                f = (rk[(int64_t)i * RB_WORDS + (a >> 5)] >> (a & 31)) & 1u;
            }
            const unsigned long long m = __ballot(f);
            if (f) {
                const int pos = nR + __popcll(m & ((1ull << lane) - 1ull));
                R[pos] = c;
                for (unsigned sl = rh_slot(c);; sl = (sl + 1) & (KR_HASH - 1)) {   // (the entries of a rank row are distinct)
                    if (atomicCAS(&Rhash[sl], -1, c) == -1) break;
                }
                atomicOr(&Emask[c >> 5], 1u << (c & 31));
            }
            nR += __popcll(m);
        }
        if (lane == 0) {
            s_nR = nR;
            if (r_count) r_count[li] = nR;   // |R(i, k1)|, summed afterwards (statistics)
        }
    }
    __syncthreads();
    const int nR = s_nR;

    // expansion: the k/2-reciprocal set of every candidate, accepted on > 2/3 overlap with R.  A half-wave (32
    // lanes) per candidate; h <= 32 entries are tested in one step, larger h in chunks of 32.
    {
        const int half = tid >> 5, hl = tid & 31;          // 8 half-waves
        const unsigned long long hmask = (hl == (lane & 31) && (lane & 32)) ? 0xffffffff00000000ull : 0x00000000ffffffffull;
        for (int a0 = 0; a0 < nR; a0 += 8) {
            const int a = a0 + half;
            const bool live = a < nR;
            const int cand = live ? R[a] : 0;
            const int *cf = rank + (int64_t)cand * KR;
            int nRc = 0, inter = 0;
            unsigned okbits = 0u;
            for (int b0 = 0, ch = 0; b0 < h; b0 += 32, ++ch) {
                const int b = b0 + hl;
                bool ok = false, inr = false;
                if (live && b < h) {
                    ok = (rh[(int64_t)cand * RB_WORDS + (b >> 5)] >> (b & 31)) & 1u;
                    if (ok) {
                        const int f = cf[b];
                        for (unsigned sl = rh_slot(f);; sl = (sl + 1) & (KR_HASH - 1)) {
                            const int hv = Rhash[sl];
                            if (hv == f) inr = true;
                            if (hv == f || hv == -1) break;
                        }
                    }
                }
                nRc += __popcll(__ballot(ok) & hmask);
                inter += __popcll(__ballot(inr) & hmask);
                if (ok) okbits |= (1u << ch);
            }
            if (live && (double)inter > (2.0 / 3.0) * (double)nRc) {
                for (int b0 = 0, ch = 0; b0 < h; b0 += 32, ++ch) {
                    const int b = b0 + hl;
                    if (b < h && ((okbits >> ch) & 1u)) {
                        const int f = cf[b];
                        atomicOr(&Emask[f >> 5], 1u << (f & 31));
                    }
                }
            }
        }
    }
    __syncthreads();

    // np.unique(expansion index) == set bits of Emask in ascending order
    if (wave == 0) {
        const int n = extract_bits_sorted(Emask, nw, Elist, lane);
        if (lane == 0) s_nE = n;
    }
    if (SPARSE)
        for (int k = tid; k < d; k += 256) qrow[k] = feat[(int64_t)i * d + k];
    __syncthreads();
    const int nE = s_nE;
    const float mx = rowmax[li];
    if (SPARSE) {
        // entries among the first K neighbours (all of R, i.e. nearly everything) have their exact distance already;
        // the others are evaluated now
        for (int t = tid; t < nE; t += 256) {
            const int j = Elist[t];
            int pos = -1;
            for (int a = 0; a < K; ++a) pos = (fwd[a] == j) ? a : pos;
            if (pos >= 0) {
                wbuf[t] = mpreid_np_expf(-__fdiv_rn(fdist[pos], mx));
            } else {
                miss[atomicAdd(&s_nmiss, 1)] = (uint16_t)t;
            }
        }
        __syncthreads();
        const int nmiss = s_nmiss;
        if (wave == 0 && nmiss > 0) {
            const float ni = norms[i];
            for (int t0 = 0; t0 < nmiss; t0 += 64) {
                const int u = t0 + lane;
                const int t = u < nmiss ? (int)miss[u] : -1;
                const float dij = wave_exact_dists(qrow, feat, d, t >= 0 ? Elist[t] : -1, ni, norms, xtile, lane);
                if (t >= 0) wbuf[t] = mpreid_np_expf(-__fdiv_rn(dij, mx));
            }
        }
    } else {
        const float *row = MT + (int64_t)li * ld;
        for (int t = tid; t < nE; t += 256) wbuf[t] = mpreid_np_expf(-__fdiv_rn(row[Elist[t]], mx));
    }
    __syncthreads();
    if (wave != 0) return;
    const float s = wave_pairwise_sum(wbuf, nE, lane);
    // V[i, E] = fp16(weight / sum); only non-zero halves are kept (V != 0 tests later)
    int out = 0;
    for (int t0 = 0; t0 < nE; t0 += 64) {
        const int t = t0 + lane;
        uint16_t hv = 0;
        if (t < nE) hv = h_from_f32(__fdiv_rn(wbuf[t], s));
        const bool nz = (hv & 0x7fffu) != 0;
        const unsigned long long m = __ballot(nz);
        if (nz) {
            const int p = out + __popcll(m & ((1ull << lane) - 1ull));
            vidx[(int64_t)li * vcap + p] = Elist[t];
            vval[(int64_t)li * vcap + p] = hv;
        }
        out += __popcll(m);
    }
    if (lane == 0) vcnt[li] = out;
}

// ---------------------------------------------------------------------------------------------
// local query expansion (utils/reranking.py:73-78): V_qe[i] = fp16( sum_{m<k2} V[rank[i][m]] / k2 )
// fp32 sum in rank order, true divide by fp32(k2).  One wave per row.
// ---------------------------------------------------------------------------------------------
// V (vcnt/vidx/vval, row stride vcap) is global; ucnt / qcnt / qidx / qval are indexed by the local row
// Walk the sparse V rows of the first k2 neighbours of row i (one wave), f(c, hv) for every entry, neighbours in rank
// order.  The neighbour indices and their entry counts are fetched lane-parallel, and the first 64 entries of EIGHT
// neighbours are requested before the first is consumed (unconditional, clamped loads: counted waits) -- the plain loop
// paid three dependent memory round trips per neighbour, 45 per row and pass (4.0 ms at N = 100 000).
template <bool WITH_VAL, typename F>
__device__ __forceinline__ void qe_walk(const int *__restrict__ rank, int64_t i, int KR, int k2, const int *__restrict__ vcnt,
                                        const int *__restrict__ vidx, const uint16_t *__restrict__ vval, int vcap, int lane,
                                        F &&f) {
    for (int mb = 0; mb < k2; mb += 64) {
        const int mm = mb + lane;
        const int my_r = mm < k2 ? rank[i * KR + mm] : 0;
        const int my_cnt = mm < k2 ? vcnt[my_r] : 0;
        const int mtop = (k2 - mb < 64) ? k2 - mb : 64;
        for (int m0 = 0; m0 < mtop; m0 += 8) {
            int c[8], cn[8];
            uint16_t hv[8];
            int64_t rbase[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int sl = (m0 + u < mtop) ? m0 + u : m0;   // (wave-uniform) past the end: a copy of neighbour m0, count 0
                const int r = __builtin_amdgcn_readlane(my_r, sl);
                cn[u] = (m0 + u < mtop) ? __builtin_amdgcn_readlane(my_cnt, sl) : 0;
                rbase[u] = (int64_t)r * vcap;
                const int a = lane < cn[u] ? lane : 0;
                c[u] = vidx[rbase[u] + a];
                if (WITH_VAL) hv[u] = vval[rbase[u] + a];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (lane < cn[u]) f(c[u], WITH_VAL ? hv[u] : (uint16_t)0);
                for (int a = lane + 64; a < cn[u]; a += 64)   // rows with more than 64 entries
                    f(vidx[rbase[u] + a], WITH_VAL ? vval[rbase[u] + a] : (uint16_t)0);
                if (WITH_VAL) __builtin_amdgcn_wave_barrier();
            }
        }
    }
}

__global__ __launch_bounds__(64) void qe_count_kernel(int64_t N, const int *__restrict__ rank, int KR, int k2,
                                                      const int *__restrict__ vcnt, const int *__restrict__ vidx,
                                                      int vcap, int *__restrict__ ucnt, int row0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned *mask = (unsigned *)smem;
    const int nw = (int)((N + 31) >> 5);
    const int lane = threadIdx.x;
    const int li = blockIdx.x;
    const int i = row0 + li;
    for (int w = lane; w < nw; w += 64) mask[w] = 0u;
    __syncthreads();
    qe_walk<false>(rank, (int64_t)i, KR, k2, vcnt, vidx, nullptr, vcap, lane,
                   [&](int c, uint16_t) { atomicOr(&mask[c >> 5], 1u << (c & 31)); });
    __syncthreads();
    int tot = 0;
    for (int w = lane; w < nw; w += 64) tot += __popc(mask[w]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) tot += __shfl_xor(tot, off, 64);
    if (lane == 0) ucnt[li] = tot;
}

__global__ __launch_bounds__(64) void qe_fill_kernel(int64_t N, const int *__restrict__ rank, int KR, int k2,
                                                     const int *__restrict__ vcnt, const int *__restrict__ vidx,
                                                     const uint16_t *__restrict__ vval, int vcap, int qcap,
                                                     int *__restrict__ qcnt, int *__restrict__ qidx,
                                                     uint16_t *__restrict__ qval, int row0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nw = (int)((N + 31) >> 5);
    unsigned *mask = (unsigned *)smem;
    int *wpre = (int *)(mask + nw);
    float *acc = (float *)(wpre + nw);
    int *ulist = (int *)(acc + qcap);
    const int lane = threadIdx.x;
    const int li = blockIdx.x;
    const int i = row0 + li;
    for (int w = lane; w < nw; w += 64) mask[w] = 0u;
    for (int t = lane; t < qcap; t += 64) acc[t] = 0.0f;
    __syncthreads();
    qe_walk<false>(rank, (int64_t)i, KR, k2, vcnt, vidx, nullptr, vcap, lane,
                   [&](int c, uint16_t) { atomicOr(&mask[c >> 5], 1u << (c & 31)); });
    __syncthreads();
    // sorted union (np.unique) and, as a by-product of the same scan, the word prefix: slot(c) = wpre[c>>5] +
    // popc(mask[c>>5] & below(c))   (one pass over the N-bit mask instead of two)
    const int nU = extract_bits_sorted(mask, nw, ulist, lane, wpre);
    __syncthreads();
    // fp32 accumulation in rank order; one writer per slot and neighbour (a one-wave workgroup: its LDS operations
    // execute in program order)
    qe_walk<true>(rank, (int64_t)i, KR, k2, vcnt, vidx, vval, vcap, lane, [&](int c, uint16_t hv) {
        const int slot = wpre[c >> 5] + __popc(mask[c >> 5] & ((1u << (c & 31)) - 1u));
        acc[slot] = acc[slot] + h_to_f32(hv);
    });
    __syncthreads();
    const float k2f = (float)k2;
    int out = 0;
    for (int t0 = 0; t0 < nU; t0 += 64) {
        const int t = t0 + lane;
        uint16_t hv = 0;
        if (t < nU) hv = h_from_f32(__fdiv_rn(acc[t], k2f));
        const bool nz = (hv & 0x7fffu) != 0;
        const unsigned long long mm = __ballot(nz);
        if (nz) {
            const int p = out + __popcll(mm & ((1ull << lane) - 1ull));
            qidx[(int64_t)li * qcap + p] = ulist[t];
            qval[(int64_t)li * qcap + p] = hv;
        }
        out += __popcll(mm);
    }
    if (lane == 0) qcnt[li] = out;
}

// ---------------------------------------------------------------------------------------------
// inverted index (utils/reranking.py:80-82): CSC of V.  Order inside a column is irrelevant to
// the result (each row occurs once per column), so the fill uses atomic cursors.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void csc_count_kernel(int64_t N, const int *__restrict__ qcnt,
                                                        const int *__restrict__ qidx, int qcap,
                                                        unsigned *__restrict__ ccnt, int64_t row_lo) {
    const int lane = threadIdx.x & 63;
    const int64_t i = row_lo + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= N) return;
    const int cnt = qcnt[i];
    for (int a = lane; a < cnt; a += 64) atomicAdd(&ccnt[qidx[i * qcap + a]], 1u);
}

// single-workgroup exclusive scan of ccnt[0..N) -> cptr[0..N]; also zeroes ccnt for reuse as cursor
__global__ __launch_bounds__(1024) void csc_scan_kernel(int64_t N, unsigned *__restrict__ ccnt,
                                                        long long *__restrict__ cptr) {
    __shared__ long long part[1024];
    const int tid = threadIdx.x;
    const int64_t per = (N + 1023) / 1024;
    const int64_t lo = (tid * per < N) ? tid * per : N, hi = (lo + per < N) ? lo + per : N;
    long long s = 0;
    for (int64_t c = lo; c < hi; ++c) s += ccnt[c];
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        long long run = 0;
        for (int t = 0; t < 1024; ++t) {
            const long long v = part[t];
            part[t] = run;
            run += v;
        }
        cptr[N] = run;
    }
    __syncthreads();
    long long run = part[tid];
    for (int64_t c = lo; c < hi; ++c) {
        cptr[c] = run;
        run += ccnt[c];
        ccnt[c] = 0u;
    }
}

__global__ __launch_bounds__(256) void csc_fill_kernel(int64_t N, const int *__restrict__ qcnt,
                                                       const int *__restrict__ qidx,
                                                       const uint16_t *__restrict__ qval, int qcap,
                                                       const long long *__restrict__ cptr,
                                                       unsigned *__restrict__ cursor, int *__restrict__ crow,
                                                       uint16_t *__restrict__ cval, int64_t row_lo) {
    const int lane = threadIdx.x & 63;
    const int64_t i = row_lo + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= N) return;
    const int cnt = qcnt[i];
    for (int a = lane; a < cnt; a += 64) {
        const int c = qidx[i * qcap + a];
        const long long p = cptr[c] + (long long)atomicAdd(&cursor[c], 1u);
        crow[p] = (int)i;
        cval[p] = qval[i * qcap + a];
    }
}

// ---------------------------------------------------------------------------------------------
// Inverted index WITHOUT global atomics (the atomic-cursor build above ran at 1 % of HBM bandwidth: 25 M device-scope
// atomics for 12.5 M entries).  Counting sort with the rows cut into CSC_B blocks:
//   1. every block histograms the columns of its rows in LDS (LDS atomics) and writes its histogram row H[b][.]
//   2. per column: exclusive scan of H[.][c] over the blocks (in place) and the column total -> ccnt[c];
//      the existing single-workgroup scan turns ccnt into cptr
//   3. every block re-walks its rows with LDS cursors cptr[c] + H[b][c] and writes (row, value) to its slots
// Columns are processed in ranges of CSC_CR (LDS: 4 bytes per column) -- one range up to N = 36 864.
// The order of the rows inside a column depends on LDS-atomic arrival order within a block; the Jaccard sum does
// not (every row occurs once per column and owns its own accumulator).
// ---------------------------------------------------------------------------------------------
constexpr int CSC_B = 256, CSC_CR = 36864;
static_assert(CSC_B == 256, "the workspace layouts reserve 256 block histograms");
__global__ __launch_bounds__(1024) void csc2_hist_kernel(int64_t N, const int *__restrict__ qcnt, const int *__restrict__ qidx,
                                                         int qcap, int rows_per_block, unsigned *__restrict__ H,
                                                         int64_t row_lo, int64_t col_lo, int64_t col_hi) {
    // columns [col_lo, col_hi) only (the whole index: 0, N; a rank's column shard of the sharded build: its range)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned *hist = (unsigned *)smem;
    const int b = blockIdx.x;
    const int64_t c0 = col_lo + (int64_t)blockIdx.y * CSC_CR;
    const int cw = (int)((col_hi - c0 < CSC_CR) ? col_hi - c0 : CSC_CR);
    for (int c = threadIdx.x; c < cw; c += 1024) hist[c] = 0u;
    __syncthreads();
    const int64_t r_lo = row_lo + (int64_t)b * rows_per_block, r_hi = (r_lo + rows_per_block < N) ? r_lo + rows_per_block : N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t i = r_lo + wave; i < r_hi; i += 16) {
        const int cnt = qcnt[i];
        for (int a = lane; a < cnt; a += 64) {
            const int64_t c = qidx[i * qcap + a] - c0;
            if (c >= 0 && c < cw) atomicAdd(&hist[c], 1u);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < cw; c += 1024) H[(int64_t)b * N + c0 + c] = hist[c];
}
// per column: H[b][c] <- sum of H[b'][c] over b' < b; ccnt[c] = total
__global__ __launch_bounds__(256) void csc2_colscan_kernel(int64_t N, unsigned *__restrict__ H, unsigned *__restrict__ ccnt,
                                                           int64_t col_lo, int64_t col_hi) {
    const int64_t c = col_lo + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= col_hi) return;
    unsigned run = 0u;
#pragma unroll 8
    for (int b = 0; b < CSC_B; ++b) {
        const unsigned v = H[(int64_t)b * N + c];
        H[(int64_t)b * N + c] = run;
        run += v;
    }
    ccnt[c] = run;
}
// cptr = exclusive scan of the column totals, in three fully parallel launches (the single-workgroup csc_scan_kernel
// above walks N / 1024 counts per thread at a stride and scans 1024 partial sums on one lane: 47 us at N = 20 000,
// 260 us at N = 100 000).  Tiles of 1024 columns.
__global__ __launch_bounds__(256) void scan_tile_sums_kernel(int64_t N, const unsigned *__restrict__ cnt,
                                                             unsigned long long *__restrict__ tile_sum) {
    __shared__ unsigned long long ws[4];
    const int64_t c0 = (int64_t)blockIdx.x * 1024 + threadIdx.x * 4;
    unsigned long long s = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) s += (c0 + u < N) ? cnt[c0 + u] : 0u;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
// tile_sum[0..nt) -> exclusive prefix in place; total -> cptr_end[0]
__global__ __launch_bounds__(1024) void scan_tile_bases_kernel(int nt, unsigned long long *__restrict__ tile_sum,
                                                               long long *__restrict__ cptr_end) {
    __shared__ unsigned long long ws[16];
    __shared__ unsigned long long carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int t0 = 0; t0 < nt; t0 += 1024) {
        const int t = t0 + tid;
        const unsigned long long v = t < nt ? tile_sum[t] : 0ull;
        unsigned long long x = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long y = __shfl_up(x, off, 64);
            if (lane >= off) x += y;
        }
        if (lane == 63) ws[wave] = x;
        __syncthreads();
        unsigned long long wbase = 0;
        for (int w = 0; w < wave; ++w) wbase += ws[w];
        const unsigned long long base = carry;
        if (t < nt) tile_sum[t] = base + wbase + x - v;
        __syncthreads();
        if (tid == 1023) carry = base + wbase + x;
        __syncthreads();
    }
    if (tid == 0) cptr_end[0] = (long long)carry;
}
// cptr[c] = tile base + exclusive prefix inside the tile; cnt is zeroed for reuse as a cursor
__global__ __launch_bounds__(256) void scan_apply_kernel(int64_t N, unsigned *__restrict__ cnt,
                                                         const unsigned long long *__restrict__ tile_base,
                                                         long long *__restrict__ cptr) {
    __shared__ unsigned long long ws[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t c0 = (int64_t)blockIdx.x * 1024 + threadIdx.x * 4;
    unsigned v[4];
    unsigned long long s = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        v[u] = (c0 + u < N) ? cnt[c0 + u] : 0u;
        s += v[u];
    }
    unsigned long long x = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long y = __shfl_up(x, off, 64);
        if (lane >= off) x += y;
    }
    if (lane == 63) ws[wave] = x;
    __syncthreads();
    unsigned long long run = tile_base[blockIdx.x] + x - s;
    for (int w = 0; w < wave; ++w) run += ws[w];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (c0 + u < N) {
            cptr[c0 + u] = (long long)run;
            cnt[c0 + u] = 0u;
        }
        run += v[u];
    }
}

// chunk-boundary table of the Jaccard stage: HB[c][k] = absolute position (in crow / cval) of the first entry of column c
// that lies in row chunk k (= row block k * bpc), HB[c][nchunks] = end of the column.  One 8-byte read per (query,
// chunk, column) from a table of a few MB instead of four scattered reads of cptr and the [256][N] histograms.
__global__ __launch_bounds__(256) void csc2_bounds_kernel(int64_t N, int nchunks, int bpc, const unsigned *__restrict__ H,
                                                          const long long *__restrict__ cptr, unsigned *__restrict__ HB,
                                                          int64_t col_lo, int64_t col_hi) {
    const int64_t c = col_lo + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= col_hi) return;
    const unsigned base = (unsigned)cptr[c];
    for (int k = 0; k < nchunks; ++k) HB[c * (nchunks + 1) + k] = base + H[(int64_t)k * bpc * N + c];
    HB[c * (nchunks + 1) + nchunks] = (unsigned)cptr[c + 1];
}
__global__ __launch_bounds__(1024) void csc2_fill_kernel(int64_t N, const int *__restrict__ qcnt, const int *__restrict__ qidx,
                                                         const uint16_t *__restrict__ qval, int qcap, int rows_per_block,
                                                         const unsigned *__restrict__ H, const long long *__restrict__ cptr,
                                                         unsigned *__restrict__ cpk, int64_t row_lo, int blocks_per_chunk,
                                                         int64_t col_lo, int64_t col_hi) {
    // entries are PACKED: (row - first row of the Jaccard chunk the row block belongs to) << 16 | fp16 bits of V_qe
    // (a chunk has at most 24 K rows): 4 bytes per entry in ONE array instead of 4 + 2 in two -- a third fewer bytes
    // and, at large N, ~40 % fewer 128-byte lines per gathered sub-range
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned *cur = (unsigned *)smem;
    const int b = blockIdx.x;
    const int64_t c0 = col_lo + (int64_t)blockIdx.y * CSC_CR;
    const int cw = (int)((col_hi - c0 < CSC_CR) ? col_hi - c0 : CSC_CR);
    for (int c = threadIdx.x; c < cw; c += 1024) cur[c] = (unsigned)cptr[c0 + c] + H[(int64_t)b * N + c0 + c];
    __syncthreads();
    const int64_t r_lo = row_lo + (int64_t)b * rows_per_block, r_hi = (r_lo + rows_per_block < N) ? r_lo + rows_per_block : N;
    const int64_t chunk_row0 = row_lo + (int64_t)(b / blocks_per_chunk) * blocks_per_chunk * rows_per_block;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t i = r_lo + wave; i < r_hi; i += 16) {
        const int cnt = qcnt[i];
        for (int a = lane; a < cnt; a += 64) {
            const int64_t c = qidx[i * qcap + a] - c0;
            if (c >= 0 && c < cw) {
                const unsigned p = atomicAdd(&cur[c], 1u);
                cpk[p] = ((unsigned)(i - chunk_row0) << 16) | (unsigned)qval[i * qcap + a];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Jaccard min-sum with fp16 accumulation in ascending column order + final blend
// (utils/reranking.py:84-100).  One 256-thread workgroup per query; t[] lives in LDS as fp16 bits,
// r-space processed in chunks of rch entries so that any N fits.
// ---------------------------------------------------------------------------------------------
// Two forms (jaccard_plan): 256 threads per (query, chunk of <= 24 K rows) with one barrier per column -- whole columns
// when N - nq <= 24 576 -- and ONE wave per (query, chunk of ~8 K rows) without barriers when the 256-thread form would need
// several chunks (N = 100 000: 45.5 -> 33.8 ms, MSMT17 shape 21.8 -> 18.6 ms, with the counted waits, the register-held
// column table and the branch-free accumulation below; the same one-wave split measured 41 / 25 ms before those).
constexpr int JT = 256; // threads per query workgroup (multi-wave form)
// NPF entries per thread and column are held in registers (columns up to NPF * JT entries; longer ones take the direct
// path below), the gathers of PD columns are in flight.  <2, 4> suits whole columns (~600 entries: 43 KB in flight per CU
// at three workgroups); at large N a workgroup sees ~130-entry sub-ranges and <1, 12> keeps as many bytes in flight.
template <int JT_, int NPF, int PD, bool PACKED>
__global__ __launch_bounds__(JT_) void jaccard_kernel(int64_t N, int64_t nq, const float *__restrict__ MT, int64_t ld,
                                                      const float *__restrict__ rowmax,
                                                      const int *__restrict__ qcnt, const int *__restrict__ qidx,
                                                      const uint16_t *__restrict__ qval, int qcap,
                                                      const long long *__restrict__ cptr,
                                                      const int *__restrict__ crow, const uint16_t *__restrict__ cval,
                                                      int rch, uint16_t one_minus_lam_h, float lam32,
                                                      float *__restrict__ out, int64_t ldo,
                                                      unsigned long long *__restrict__ pair_counter, int q0,
                                                      const unsigned *__restrict__ H, int rows_per_block,
                                                      int blocks_per_chunk, int dbg) {
    // H != NULL (inverted index built by csc2_*: the entries of a column are grouped by row block; H = the chunk
    // boundaries of every column): blockIdx.y selects a chunk of blocks_per_chunk row blocks = rch rows, and the
    // workgroup gathers exactly the sub-range of every column that lies in it.  (Round 1 walked the r-space chunk by chunk inside one workgroup and re-read every column in
    // full for every chunk: 3x the gather traffic at N = 100 000, at one workgroup per CU.)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t *t = (uint16_t *)smem;                                   // [rch]
    long long *cp0 = (long long *)(smem + align_up((size_t)rch * 2 + 16, 16)); // [cnt]  (t[rch] = dummy slot)
    int *clen = (int *)(cp0 + qcap);                                  // [cnt]
    uint16_t *vi = (uint16_t *)(clen + qcap);                         // [cnt]
    const int tid = threadIdx.x;
    const int64_t i = blockIdx.x;              // local query row: MT / rowmax / out are indexed by it
    const int64_t ig = (int64_t)q0 + i;        // global row: the sparse V rows are indexed by it
    const int cnt = qcnt[ig];
    unsigned long long pairs = 0;
    const int b_lo = H ? (int)blockIdx.y * blocks_per_chunk : 0;
    const int b_hi = b_lo + blocks_per_chunk;   // >= CSC_B for the last chunk
    const int nb1 = (int)gridDim.y + 1;
    for (int a = tid; a < cnt; a += JT_) {
        const int c = qidx[ig * qcap + a];
        long long p0, p1;
        if (H) {   // H = chunk-boundary table [N][nchunks + 1] (csc2_bounds_kernel)
            p0 = H[(int64_t)c * nb1 + blockIdx.y];
            p1 = H[(int64_t)c * nb1 + blockIdx.y + 1];
        } else {
            p0 = cptr[c];
            p1 = cptr[c + 1];
        }
        cp0[a] = p0;
        clen[a] = (int)(p1 - p0);
        vi[a] = qval[ig * qcap + a];
        pairs += (unsigned long long)(p1 - p0);
    }
    if (pair_counter) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) pairs += __shfl_xor(pairs, off, 64);
        if ((tid & 63) == 0 && pairs) atomicAdd(pair_counter, pairs);
    }
    const float mx = rowmax[i];
    const float *row = MT + i * ld;
    const uint16_t H1 = 0x3c00u, H2 = 0x4000u;
    const unsigned long long last = cnt > 0 ? (unsigned long long)(cptr[N] - 1) : 0ull;   // cnt > 0: the index is not empty
    // the inverted index holds the rows [nq, N) only (launch_csc): the accumulators start at row nq
    const int64_t r_first = nq + (H ? (int64_t)b_lo * rows_per_block : 0);
    const int64_t r_end = nq + (int64_t)b_hi * rows_per_block;
    const int64_t r_last = H ? (r_end < N ? r_end : N) : N;   // (rch, the LDS size, may be rounded up past it)
    for (int64_t r0 = r_first; r0 < r_last; r0 += rch) {
        const int64_t r1 = (r0 + rch < r_last) ? r0 + rch : r_last;
        for (int r = tid; r < rch; r += JT_) t[r] = 0;
        __syncthreads();
        const int r0i = (int)r0;
        const unsigned span = (unsigned)(r1 - r0);
        // ascending column; rows of one column are distinct, so the threads of a column never collide and
        // one barrier per column keeps the fp16 accumulation order of the reference.  The (row, value)
        // pairs of column a+1 are requested into registers before column a is applied, so the L2/HBM
        // latency of the gathers is paid once per pipeline fill instead of once per column.
        // the (row, value) pairs of columns a+1 .. a+PD are requested before column a is applied (one column ahead
        // left ~1.6 us per column exposed: a barrier plus most of an L2 / HBM round trip)
        // Every gather is UNCONDITIONAL (clamped column index, entry 0 for the lanes past the column's end) and a
        // column's registers are refilled only after it has been applied: hipcc then counts the loads in flight exactly
        // and waits with vmcnt((PD - 1) * 2 * NPF).  (With the loads under `if (e < len)` / `if (a + PD < cnt)` it could
        // not know how many younger loads exist, and with the refill issued before the apply it rotated the registers
        // by moves at the loop end: both forms drained to vmcnt(0) once per PD columns -- a full HBM round trip per
        // group, ~0.8 us per column.)
        // The column table (start, length, V[i][c]) is NOT read from LDS per column -- four dependent LDS round trips per
        // step were most of a step's time at one or two waves per SIMD -- but once per 64 columns: lane l keeps the
        // entries of column 64 b + l in registers (one set for the column being applied, one for the column being
        // prefetched, PD columns ahead) and a step picks its column with v_readlane.
        static_assert(64 % PD == 0, "batch boundaries must fall on group boundaries");
        int pr[PD][NPF];
        uint16_t pv[PD][NPF];
        const int lane = tid & 63;
        unsigned f_lo = 0, f_hi = 0;
        int f_len = 0, a_len = 0, a_vi = 0;
        auto load_fetch_batch = [&](int base) {
            int idx = base + lane;
            idx = idx < cnt ? idx : cnt - 1;
            const unsigned long long p0 = (unsigned long long)cp0[idx];
            f_lo = (unsigned)p0;
            f_hi = (unsigned)(p0 >> 32);
            f_len = clen[idx];
        };
        auto load_apply_batch = [&](int base) {
            int idx = base + lane;
            idx = idx < cnt ? idx : cnt - 1;
            a_len = clen[idx];
            a_vi = vi[idx];
        };
        // (the loads use a wave-uniform base + a 32-bit lane offset; lanes past the column's end read its first entry
        // -- `last` keeps that in bounds for an empty column at the very end of the index -- and are masked when applied)
        auto fetch = [&](int a, int (&er)[NPF], uint16_t (&ev)[NPF]) {   // a >= cnt: the batch entry is a clamped copy
            const int sl = a & 63;
            unsigned long long p0 = (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)f_lo, sl) |
                                    ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)f_hi, sl) << 32);
            p0 = p0 < last ? p0 : last;
            const int len = __builtin_amdgcn_readlane(f_len, sl);
            const int *cb = crow + p0;
            const uint16_t *vb = cval + p0;
#pragma unroll
            for (int k = 0; k < NPF; ++k) {
                const unsigned e = (unsigned)(tid + k * JT_);
                const unsigned off = e < (unsigned)len ? e : 0u;
                er[k] = cb[off];
                if (!PACKED) ev[k] = vb[off];
            }
        };
        if (cnt > 0) {
            load_fetch_batch(0);
#pragma unroll
            for (int d = 0; d < PD; ++d) fetch(d, pr[d], pv[d]);
        }
        for (int a0 = 0; a0 < cnt; a0 += PD) {
            if ((a0 & 63) == 0) load_apply_batch(a0);
            if (((a0 + PD) & 63) == 0) load_fetch_batch(a0 + PD);
#pragma unroll
            for (int d = 0; d < PD; ++d) {
                const int a = a0 + d;
                if (a < cnt) {
                    const uint16_t vic = (uint16_t)__builtin_amdgcn_readlane(a_vi, a & 63);
                    const int len = __builtin_amdgcn_readlane(a_len, a & 63);
                    // branch free: lanes without an entry in this chunk add into a dummy slot behind the accumulators
                    // (the NPF entries of a lane belong to one column, i.e. to different rows: read all accumulators,
                    // then add, then write -- one LDS round trip per column instead of NPF)
                    unsigned idx[NPF];
                    uint16_t tv[NPF];
#pragma unroll
                    for (int k = 0; k < NPF; ++k) {
                        // PACKED: chunk-relative row in the high half (every entry of the sub-range lies in the chunk)
                        const unsigned rl = PACKED ? ((unsigned)pr[d][k] >> 16) : (unsigned)(pr[d][k] - r0i);
                        const bool ok = (tid + k * JT_ < len) && (PACKED || rl < span) && !(dbg & 1);
                        idx[k] = ok ? rl : (unsigned)(rch + k);   // NPF dummy slots
                        tv[k] = t[idx[k]];
                    }
#pragma unroll
                    for (int k = 0; k < NPF; ++k) {
                        const uint16_t vv = PACKED ? (uint16_t)((unsigned)pr[d][k] & 0xffffu) : pv[d][k];
                        t[idx[k]] = h_add_native(tv[k], mpreid_h_min_nonneg(vic, vv));
                    }
                    if (len > NPF * JT_ && !(dbg & 8)) { // rare long column: the tail is gathered directly
                        const long long p0 = cp0[a];
                        for (int e = tid + NPF * JT_; e < len; e += JT_) {
                            if (PACKED) {
                                const unsigned pe = (unsigned)crow[p0 + e];
                                const uint16_t m = mpreid_h_min_nonneg(vic, (uint16_t)(pe & 0xffffu));
                                t[pe >> 16] = h_add_native(t[pe >> 16], m);
                            } else {
                                const int r = crow[p0 + e];
                                if (r >= r0 && r < r1) {
                                    const uint16_t m = mpreid_h_min_nonneg(vic, cval[p0 + e]);
                                    t[r - r0] = h_add_native(t[r - r0], m);
                                }
                            }
                        }
                    }
                }
                fetch(a + PD, pr[d], pv[d]);
                __syncthreads();
            }
        }
        const int64_t jlo = (r0 > nq) ? r0 : nq;
        // eight independent loads per thread and trip (a one-wave workgroup would otherwise pay a memory round trip per
        // 64 elements)
        for (int64_t jb = jlo; jb < ((dbg & 4) ? jlo : r1); jb += 8 * JT_) {
            float dv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t j = jb + u * JT_ + tid;
                dv[u] = row[j < r1 ? j : r1 - 1];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t j = jb + u * JT_ + tid;
                if (j < r1) {
                    const uint16_t tv = t[j - r0];
                    const uint16_t den = h_sub_native(H2, tv);
                    const uint16_t qt = h_div_native(tv, den);
                    const uint16_t jac = h_sub_native(H1, qt);
                    const uint16_t jl = h_mul_native(jac, one_minus_lam_h);
                    const float o = __fdiv_rn(dv[u], mx);
                    out[i * ldo + (j - nq)] = h_to_f32(jl) + o * lam32;
                }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// One-wave form of the Jaccard stage WITHOUT a column table in LDS (blocked + packed index only): the workgroup's LDS is
// the fp16 accumulators of its row chunk and nothing else (16 KB at 8 K rows: nine workgroups per CU instead of six), and
// the per-column data (start of the sub-range, its length, V[i][c]) live in lane registers, 64 columns at a time:
//   cur  = the batch the column being applied is in, nxt = the following batch (the prefetch stream, PD columns ahead,
//   may already be there), both fully resolved (position / length from the chunk-boundary table HB);
//   c2 / v2 = column indices and values of the batch after that, loaded one batch early so that resolving them at the
//   next rotation is one round trip that nothing waits for.
// In-kernel stamps of the table form at N = 100 000 (per (query, chunk) workgroup, cycles): table set-up 29 k, zeroing
// 9 k, column loop 325 k (489 per column at 1.5 waves per SIMD), output pass 69 k.
// ---------------------------------------------------------------------------------------------
template <int JT_, int NPF, int PD>
__global__ __launch_bounds__(JT_) void jaccard_wave_kernel(int64_t N, int64_t nq, const float *__restrict__ MT, int64_t ld,
                                                          const float *__restrict__ rowmax, const int *__restrict__ qcnt,
                                                          const int *__restrict__ qidx, const uint16_t *__restrict__ qval,
                                                          int qcap, const long long *__restrict__ cptr,
                                                          const unsigned *__restrict__ cpk, int rch,
                                                          uint16_t one_minus_lam_h, float lam32, float *__restrict__ out,
                                                          int64_t ldo, unsigned long long *__restrict__ pair_counter, int q0,
                                                          const unsigned *__restrict__ HB, int rows_per_block,
                                                          int blocks_per_chunk) {
    static_assert(64 % PD == 0, "batch boundaries must fall on group boundaries");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t *t = (uint16_t *)smem;            // [rch] + NPF dummy slots
    const int tid = threadIdx.x, lane = tid & 63;
    constexpr bool MULTI = JT_ > 64;   // several waves: one barrier per column keeps the accumulation order
    const int64_t i = blockIdx.x;              // local query row: MT / rowmax / out are indexed by it
    const int64_t ig = (int64_t)q0 + i;        // global row: the sparse V rows are indexed by it
    const int y = (int)blockIdx.y, nb1 = (int)gridDim.y + 1;
    const int cnt = qcnt[ig];
    const int64_t r0 = nq + (int64_t)y * blocks_per_chunk * rows_per_block;
    const int64_t r_end = r0 + (int64_t)blocks_per_chunk * rows_per_block;
    const int64_t r1 = r_end < N ? r_end : N;
    for (int r = tid; r < rch + 8; r += JT_) t[r] = 0;
    if (MULTI) __syncthreads();
    const unsigned last = cnt > 0 ? (unsigned)(cptr[N] - 1) : 0u;   // cnt > 0: the index is not empty
    unsigned pairs = 0;
    // column indices / values of a batch (lanes past the end: a copy of the last column, marked by v >> 31)
    auto load_cols = [&](int base, int &c, unsigned &v) {
        const int idx = base + lane;
        const int ic = idx < cnt ? idx : cnt - 1;
        c = qidx[ig * qcap + ic];
        v = (unsigned)qval[ig * qcap + ic] | (idx < cnt ? 0u : 0x80000000u);
    };
    // start (clamped into the index) and (length << 16 | V[i][c]) of the chunk's sub-range of every column of a batch
    auto resolve = [&](int c, unsigned v, unsigned &p0, unsigned &lenvi) {
        const unsigned a = HB[(int64_t)c * nb1 + y], b = HB[(int64_t)c * nb1 + y + 1];
        const unsigned len = (v >> 31) ? 0u : b - a;
        p0 = a < last ? a : last;
        lenvi = (len << 16) | (v & 0xffffu);
        pairs += len;
    };
    if (cnt > 0) {
        unsigned cur_p0, cur_lv, nxt_p0, nxt_lv, v2;
        int c2;
        {
            int c0, c1;
            unsigned v0, v1;
            load_cols(0, c0, v0);
            load_cols(64, c1, v1);
            load_cols(128, c2, v2);
            resolve(c0, v0, cur_p0, cur_lv);
            resolve(c1, v1, nxt_p0, nxt_lv);
        }
        unsigned pr[PD][NPF];
        // (wave-uniform base + 32-bit lane offset; lanes past the column's end read its first entry and are masked when
        // applied; every gather is unconditional so that hipcc counts the loads in flight exactly)
        auto fetch = [&](unsigned p0v, unsigned lvv, int a, unsigned (&er)[NPF]) {
            const int sl = a & 63;
            const unsigned p0 = (unsigned)__builtin_amdgcn_readlane((int)p0v, sl);
            const unsigned len = (unsigned)__builtin_amdgcn_readlane((int)lvv, sl) >> 16;
            const unsigned *cb = cpk + p0;
#pragma unroll
            for (int k = 0; k < NPF; ++k) {
                const unsigned e = (unsigned)(tid + k * JT_);
                er[k] = cb[e < len ? e : 0u];
            }
        };
#pragma unroll
        for (int d = 0; d < PD; ++d) fetch(cur_p0, cur_lv, d, pr[d]);
        for (int a0 = 0; a0 < cnt; a0 += PD) {
            if ((a0 & 63) == 0 && a0 > 0) {   // the apply stream enters the next batch
                cur_p0 = nxt_p0;
                cur_lv = nxt_lv;
                resolve(c2, v2, nxt_p0, nxt_lv);
                load_cols(a0 + 128, c2, v2);
            }
            // the prefetch stream (columns a0 + PD ...) is in the next batch during the last group of a batch
            const bool f_nxt = ((a0 + PD) >> 6) != (a0 >> 6);
            const unsigned f_p0 = f_nxt ? nxt_p0 : cur_p0, f_lv = f_nxt ? nxt_lv : cur_lv;
#pragma unroll
            for (int d = 0; d < PD; ++d) {
                const int a = a0 + d;
                if (a < cnt) {
                    const unsigned lv = (unsigned)__builtin_amdgcn_readlane((int)cur_lv, a & 63);
                    const uint16_t vic = (uint16_t)(lv & 0xffffu);
                    const unsigned len = lv >> 16;
                    // branch free: lanes without an entry add into a dummy slot behind the accumulators; the NPF entries of
                    // a lane belong to one column, i.e. to different rows: read all accumulators, add, write
                    unsigned idx[NPF];
                    uint16_t tv[NPF];
#pragma unroll
                    for (int k = 0; k < NPF; ++k) {
                        idx[k] = ((unsigned)(tid + k * JT_) < len) ? (pr[d][k] >> 16) : (unsigned)(rch + k);
                        tv[k] = t[idx[k]];
                    }
#pragma unroll
                    for (int k = 0; k < NPF; ++k)
                        t[idx[k]] = h_add_native(tv[k], mpreid_h_min_nonneg(vic, (uint16_t)(pr[d][k] & 0xffffu)));
                    if (len > (unsigned)(NPF * JT_)) {   // long sub-range: the tail is gathered directly
                        const unsigned p0 = (unsigned)__builtin_amdgcn_readlane((int)cur_p0, a & 63);
                        for (unsigned e = (unsigned)(tid + NPF * JT_); e < len; e += JT_) {
                            const unsigned pe = cpk[p0 + e];
                            t[pe >> 16] = h_add_native(t[pe >> 16], mpreid_h_min_nonneg(vic, (uint16_t)(pe & 0xffffu)));
                        }
                    }
                }
                fetch(f_p0, f_lv, a + PD, pr[d]);
                if (MULTI) __syncthreads();
            }
        }
    }
    if (MULTI) __syncthreads();
    if (pair_counter && tid < 64) {   // (every wave resolves the same columns: wave 0 reports)
        unsigned long long ps = pairs;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) ps += __shfl_xor(ps, off, 64);
        if (lane == 0 && ps) atomicAdd(pair_counter, ps);
    }
    // blend: final = fp32(fp16(J * fp16(1 - lambda))) + O[i][j] * lambda for the chunk's gallery columns
    const float mx = rowmax[i];
    const float *row = MT + i * ld;
    const uint16_t H1 = 0x3c00u, H2 = 0x4000u;
    constexpr int OU = 16;   // independent loads per thread and trip
    for (int64_t jb = r0; jb < r1; jb += OU * JT_) {
        float dv[OU];
#pragma unroll
        for (int u = 0; u < OU; ++u) {
            const int64_t j = jb + u * JT_ + tid;
            dv[u] = row[j < r1 ? j : r1 - 1];
        }
#pragma unroll
        for (int u = 0; u < OU; ++u) {
            const int64_t j = jb + u * JT_ + tid;
            if (j < r1) {
                const uint16_t tv = t[j - r0];
                const uint16_t den = h_sub_native(H2, tv);
                const uint16_t qt = h_div_native(tv, den);
                const uint16_t jac = h_sub_native(H1, qt);
                const uint16_t jl = h_mul_native(jac, one_minus_lam_h);
                const float o = __fdiv_rn(dv[u], mx);
                out[i * ldo + (j - nq)] = h_to_f32(jl) + o * lam32;
            }
        }
    }
}

// =============================================================================================
// Candidate pipeline ("sparse" algorithm): initial_rank and the column maxima WITHOUT the N x N matrix.
//
//   1. feat -> fp16 (one rounding per element); every 16th row forms a sample.
//   2. sample pass: d~ = one-pass fp16 distances of every row against the sample ([N][N/16] fp32, 1/16 of N x N);
//      per row the r-th smallest and the largest sample value give two thresholds (rr2_threshold_kernel).
//   3. fused pass: the full symmetric fp16 GEMM on the matrix cores with the GE_CAND epilogue (gemm_f16.hip):
//      nothing is stored, entries with d~ <= tlo[i] or d~ >= thi[i] are appended to row i's candidate lists
//      (~160 + ~16 of N entries).
//   4. refinement (rr2_refine_kernel): the EXACT distance (exact_dist_chain = the oracle's fmaf chain) of the few
//      candidates that can still be among the first KR neighbours / be the row maximum, then the exact selection by
//      (O value, index).  Everything rests on one bound:   |d~ - d_exact| <= eps_i   (derivation at rr2_eps).
//      A row whose candidate list cannot be PROVEN to contain the answer (too few / too many candidates, threshold
//      closer than 2 eps to the KR-th candidate) is handed to the fallback: its whole distance row is computed
//      exactly and selected by the dense kernel.  So the result is always the dense algorithm's, bit for bit;
//      the thresholds only decide how much work that takes.
//   5. k-reciprocal expansion with on-the-fly exact distances (krecip_kernel<true>), query expansion, inverted index
//      as before; the Jaccard stage reads the exact distance rows of the QUERIES only ([nq][N]).
// HBM traffic: N x N/16 fp32 once, instead of N x N fp32 written once and read ~5 times.
// =============================================================================================

// |fp16 one-pass distance - exact fp32-chain distance| for rows i, j with norms a = |f_i|, b = |f_j|, dimension D:
//   inputs rounded to fp16: relative 2^-11 each  ->  |dot~ - dot| <= (2 * 2^-11 + 2^-22) * sum |x_k y_k| <= 2^-10 * 1.001 * a * b
//   fp16 subnormals (|x_k| < 2^-14): absolute 2^-25 per element  ->  <= 2^-25 * sqrt(D) * (a + b)
//   fp32 accumulation, either side, any order: <= D * 2^-24 * a * b each
//   d = fmaf(-2, dot, a^2 + b^2): the dot errors double; the norm sum and the final rounding add 2^-23 * (a^2 + b^2 + 2ab)
// eps_i uses b <= G = the largest row norm.
__device__ __forceinline__ float rr2_eps(float a, float G, int D) {
    const float ab = a * G;
    return ab * (0.0019551f + 2.4e-7f * (float)D) + 1.2e-7f * sqrtf((float)D) * (a + G) + 2.4e-7f * (a + G) * (a + G);
}

// rows r*stride of x (fp32 [n][d]) -> fp16 [n_out_pad][d_pad], zero padded; norms gathered alongside (padded with 0)
__global__ __launch_bounds__(256) void rr2_cast_rows_kernel(const float *__restrict__ x, const float *__restrict__ sqn,
                                                            int64_t n, int d, int stride, _Float16 *__restrict__ y,
                                                            int d_pad, float *__restrict__ sqn_out) {
    const int64_t r = blockIdx.x, src = r * stride;
    const bool ok = src < n;
    for (int k = threadIdx.x; k < d_pad; k += 256) y[r * (int64_t)d_pad + k] = (ok && k < d) ? (_Float16)x[src * d + k] : (_Float16)0.0f;
    if (sqn_out && threadIdx.x == 0) sqn_out[r] = ok ? sqn[src] : 0.0f;
}

// G = max row norm -> gstat[0]; gstat[1] = 1 if any norm is too large for fp16 operands (or not finite)
__global__ __launch_bounds__(1024) void rr2_norm_stats_kernel(const float *__restrict__ sqn, int64_t n, float *__restrict__ gstat) {
    __shared__ float red[16];
    __shared__ int bad_any;
    if (threadIdx.x == 0) bad_any = 0;
    __syncthreads();
    float m = 0.0f;
    int bad = 0;   // fmaxf DROPS a NaN operand, so non-finite norms are tracked explicitly (a NaN row must not pass)
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const float v = sqn[i];
        bad |= !(v >= 0.0f && v < 3.0e38f);
        m = fmaxf(m, v);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if (bad) atomicOr(&bad_any, 1);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) m = fmaxf(m, red[w]);
        const float G = sqrtf(m);
        gstat[0] = G;
        gstat[1] = (G < 3.0e4f && !bad_any) ? 0.0f : 1.0f;
    }
}

// per row: thresholds from the sample distances.  tlo = r-th smallest sample value (about 16 r of the N entries of
// the row lie below it), thi = sample maximum - 2 eps.  Padded rows get -inf / +inf.  256 threads per row.
__global__ __launch_bounds__(256) void rr2_threshold_kernel(const float *__restrict__ S, int64_t ldS, int ns, int64_t N,
                                                            int rsel, const float *__restrict__ sqn,
                                                            const float *__restrict__ gstat, int D,
                                                            float *__restrict__ tlo, float *__restrict__ thi,
                                                            float *__restrict__ eps, unsigned *__restrict__ cnt_lo,
                                                            unsigned *__restrict__ cnt_hi) {
    __shared__ unsigned keys[512];
    __shared__ float s_red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t i = blockIdx.x;
    if (tid == 0) {
        cnt_lo[i] = 0u;
        cnt_hi[i] = 0u;
    }
    if (i >= N) {
        if (tid == 0) {
            tlo[i] = -__builtin_huge_valf();
            thi[i] = __builtin_huge_valf();
            eps[i] = 0.0f;
        }
        return;
    }
    const float *row = S + i * ldS;
    // A threshold is all this has to produce (correctness never depends on it), so the selection is approximate and
    // cheap: every thread keeps the minimum of its strided slice, each wave sorts its 64 minima with a shuffle-only
    // bitonic network, and the r-th smallest of the four waves' r smallest is taken.  It is >= the row's true r-th
    // smallest sample value (two of the r smallest in one thread's slice hide one of them), which only adds candidates.
    unsigned k0 = 0xffffffffu;
    float mx = -3.402823466e+38f;
    for (int j = tid; j < ns; j += 256) {
        const float v = row[j];
        mx = fmaxf(mx, v);
        const unsigned k = fkey(v);
        k0 = k < k0 ? k : k0;
    }
#pragma unroll
    for (int size = 2; size <= 64; size <<= 1)
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const unsigned y = __shfl_xor(k0, stride, 64);
            const bool keep_min = (((lane & stride) == 0) == ((lane & size) == 0));
            k0 = keep_min ? (k0 < y ? k0 : y) : (k0 > y ? k0 : y);
        }
    if (lane < 16) keys[wave * 16 + lane] = k0;          // the wave's 16 smallest minima, ascending
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    if (lane == 0) s_red[wave] = mx;
    __syncthreads();
    if (wave == 0) {
        unsigned x = keys[lane];                          // 4 x 16 keys
#pragma unroll
        for (int size = 2; size <= 64; size <<= 1)
#pragma unroll
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                const unsigned y = __shfl_xor(x, stride, 64);
                const bool keep_min = (((lane & stride) == 0) == ((lane & size) == 0));
                x = keep_min ? (x < y ? x : y) : (x > y ? x : y);
            }
        keys[lane] = x;
    }
    __syncthreads();
    if (tid == 0) {
        int r = rsel < ns ? rsel : ns;
        r = r < 16 ? r : 16;
        const unsigned kk = keys[r - 1];
        const unsigned u = (kk & 0x80000000u) ? (kk & 0x7fffffffu) : ~kk;   // inverse of fkey
        const float e = rr2_eps(sqrtf(sqn[i]), gstat[0], D);
        tlo[i] = __uint_as_float(u);
        thi[i] = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3])) - 2.0f * e;
        eps[i] = e;
    }
}

// Refinement of one row (256 threads): see the header of this section.  status[i]: 0 = certified, 1 = fallback.
constexpr int RR2_MAXE = 128;   // exact evaluations per row on the `lo` side
constexpr int RR2_MAXH = 32;    // ... and on the `hi` side
template <int CAP_LO, int CAP_HI>
__global__ __launch_bounds__(256) void rr2_refine_kernel(const float *__restrict__ feat, const float *__restrict__ sqn,
                                                         int64_t N, int d, int KR, const unsigned *__restrict__ cnt_lo,
                                                         const uint2 *__restrict__ list_lo,
                                                         const unsigned *__restrict__ cnt_hi,
                                                         const uint2 *__restrict__ list_hi,
                                                         const float *__restrict__ tlo, const float *__restrict__ eps,
                                                         float *__restrict__ rowmax, int *__restrict__ rank,
                                                         float *__restrict__ rankd,
                                                         unsigned *__restrict__ fb_count, int *__restrict__ fb_rows,
                                                         unsigned fb_max, int64_t row0) {
    // lists, thresholds, rank / rankd / rowmax and the fallback list are indexed by the LOCAL row (blockIdx.x); feat
    // and sqn are the global arrays and row0 the global index of local row 0 (0 unless the rows are sharded)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    static_assert(CAP_LO <= 512 && CAP_HI <= 256, "list capacities");
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(smem);   // [512]
    unsigned long long *ekey = keys + 512;                                     // [RR2_MAXE] exact (O key, index)
    int *hsel = reinterpret_cast<int *>(ekey + RR2_MAXE);                      // [RR2_MAXH]
    float *hval = reinterpret_cast<float *>(hsel + RR2_MAXH);                  // [RR2_MAXH]
    float *qrow = hval + RR2_MAXH;                                             // [d rounded up to 4]
    float *xtile = qrow + ((d + 3) & ~3);                                      // [2 waves][64][WXD_STRIDE]
    __shared__ float s_red[4];
    __shared__ unsigned s_cnt;
    __shared__ int s_fail;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t i = blockIdx.x;
    const unsigned nlo = cnt_lo[i], nhi = cnt_hi[i];
    const float e = eps[i];
    if (tid == 0) {
        s_cnt = 0u;
        s_fail = (nlo > (unsigned)CAP_LO || nhi > (unsigned)CAP_HI || nlo < (unsigned)KR || nhi < 1u) ? 1 : 0;
    }
    for (int k = tid; k < d; k += 256) qrow[k] = feat[(row0 + i) * d + k];
    // ---- lo side: sort the candidates by approximate distance ----
    for (int t = tid; t < 512; t += 256) {
        unsigned long long kv = ~0ull;
        if (t < (int)nlo && t < CAP_LO) {
            const uint2 c = list_lo[i * CAP_LO + t];
            kv = ((unsigned long long)fkey(__uint_as_float(c.y)) << 32) | c.x;
        }
        keys[t] = kv;
    }
    __syncthreads();
    const bool fail0 = s_fail != 0;
    if (!fail0) {
        const int sort_n = nlo <= 256u ? 256 : 512;   // (most rows have ~160 candidates)
        for (int size = 2; size <= sort_n; size <<= 1)
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int a_i = tid + h * 256, partner = a_i ^ stride;
                    if (a_i < sort_n && partner > a_i) {
                        const unsigned long long a = keys[a_i], b = keys[partner];
                        const bool up = ((a_i & size) == 0);
                        if ((a > b) == up) {
                            keys[a_i] = b;
                            keys[partner] = a;
                        }
                    }
                }
                __syncthreads();
            }
    }
    // x~ = KR-th smallest approximate distance; every candidate up to x~ + 2 eps is evaluated exactly
    float xt = 0.0f, cut = 0.0f;
    int m = 0;
    if (!fail0) {
        const unsigned kx = (unsigned)(keys[KR - 1] >> 32);
        xt = __uint_as_float((kx & 0x80000000u) ? (kx & 0x7fffffffu) : ~kx);
        cut = xt + 2.0f * e;
        const unsigned kcut = fkey(cut);
        // count of keys <= cut (keys sorted ascending; nlo <= 512)
        int c = 0;
        for (int t = tid; t < (int)nlo; t += 256) c += ((unsigned)(keys[t] >> 32) <= kcut) ? 1 : 0;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
        if (lane == 0) atomicAdd(&s_cnt, (unsigned)c);
    }
    __syncthreads();
    m = (int)s_cnt;
    __syncthreads();
    // certification 1: everything that was NOT a candidate has d~ > tlo, so exact > tlo - eps; it must lie above the
    // exact KR-th, which is <= x~ + eps  ->  x~ + 2 eps <= tlo;  and the evaluation budget
    if (tid == 0) {
        if (!fail0 && (m > RR2_MAXE || !(xt + 2.0f * e <= tlo[i]))) s_fail = 1;
        s_cnt = 0u;
    }
    // ---- hi side: approximate maximum, everything within 2 eps of it is evaluated exactly ----
    float hm = -3.402823466e+38f;
    if (!fail0)
        for (int t = tid; t < (int)nhi; t += 256) hm = fmaxf(hm, __uint_as_float(list_hi[i * CAP_HI + t].y));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) hm = fmaxf(hm, __shfl_xor(hm, off, 64));
    if (lane == 0) s_red[wave] = hm;
    __syncthreads();
    hm = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    if (!fail0)
        for (int t = tid; t < (int)nhi; t += 256) {
            const uint2 c = list_hi[i * CAP_HI + t];
            if (__uint_as_float(c.y) >= hm - 2.0f * e) {
                const unsigned p = atomicAdd(&s_cnt, 1u);
                if (p < (unsigned)RR2_MAXH) hsel[p] = (int)c.x;
            }
        }
    __syncthreads();
    const int mh = (int)s_cnt;
    if (tid == 0 && (mh > RR2_MAXH || m + mh > RR2_MAXE)) s_fail = 1;   // lo and hi evaluations share 128 lanes
    __syncthreads();
    if (s_fail) {
        if (tid == 0) {
            const unsigned p = atomicAdd(fb_count, 1u);
            if (p < fb_max) fb_rows[p] = (int)i;
        }
        return;
    }
    // ---- exact distances: threads 0..m-1 the lo candidates, threads m..m+mh-1 the hi candidates (two waves) ----
    const float ni = sqn[row0 + i];
    float dex = 0.0f;
    int jx = -1;
    if (tid < m) jx = (int)(unsigned)(keys[tid] & 0xffffffffull);
    else if (tid < m + mh) jx = hsel[tid - m];
    if (wave < 2 && wave * 64 < m + mh)   // wave-uniform
        dex = wave_exact_dists(qrow, feat, d, jx, ni, sqn, xtile + wave * 64 * WXD_STRIDE, lane);
    if (tid >= m && tid < m + mh) hval[tid - m] = dex;
    __syncthreads();
    float mx = -3.402823466e+38f;
    for (int t = 0; t < mh; ++t) mx = fmaxf(mx, hval[t]);   // exact row maximum (see the header: always certified)
    // ---- exact selection by (O value, index) ----
    if (tid < RR2_MAXE) ekey[tid] = (tid < m) ? (((unsigned long long)fkey(__fdiv_rn(dex, mx)) << 32) | (unsigned)jx) : ~0ull;
    // the approximate keys are done with (every thread has read its jx before the barrier above): their memory now
    // holds (index, exact distance) by slot, for the distances of the selected neighbours written out below
    int *djx = reinterpret_cast<int *>(keys);
    float *dlo = reinterpret_cast<float *>(keys + 64);
    if (tid < m) {
        djx[tid] = jx;
        dlo[tid] = dex;
    }
    __syncthreads();
    for (int size = 2; size <= RR2_MAXE; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (tid < RR2_MAXE) {
                const int partner = tid ^ stride;
                if (partner > tid) {
                    const unsigned long long a = ekey[tid], b = ekey[partner];
                    const bool up = ((tid & size) == 0);
                    if ((a > b) == up) {
                        ekey[tid] = b;
                        ekey[partner] = a;
                    }
                }
            }
            __syncthreads();
        }
    // certification 2 (ties after the division): whatever was not evaluated has exact distance > x~ + eps (the
    // unevaluated candidates) or > tlo - eps >= x~ + eps (the rest); its O value is >= fdiv(x~ + eps, mx), which
    // must be STRICTLY above the KR-th selected O value
    const unsigned okx = (unsigned)(ekey[KR - 1] >> 32);
    const bool ok = fkey(__fdiv_rn(xt + e, mx)) > okx;
    if (!ok) {
        if (tid == 0) {
            const unsigned p = atomicAdd(fb_count, 1u);
            if (p < fb_max) fb_rows[p] = (int)i;
        }
        return;
    }
    if (tid < KR) {
        const int j = (int)(unsigned)(ekey[tid] & 0xffffffffull);
        rank[i * KR + tid] = j;
        float dj = 0.0f;   // D[i][j], exact: the expansion kernel weighs the neighbours with exp(-D / rowmax)
        for (int sl = 0; sl < m; ++sl) dj = (djx[sl] == j) ? dlo[sl] : dj;
        rankd[i * KR + tid] = dj;
    }
    if (tid == 0) rowmax[i] = mx;
}

// fallback rows: gather their features / norms, and scatter their dense results back
__global__ __launch_bounds__(256) void rr2_fb_gather_kernel(const float *__restrict__ feat, const float *__restrict__ sqn, int d,
                                                            const int *__restrict__ fb_rows,
                                                            const unsigned *__restrict__ fb_count, unsigned fb_max,
                                                            float *__restrict__ fb_feat, float *__restrict__ fb_sqn,
                                                            int64_t row0) {
    const unsigned b = blockIdx.x;
    const unsigned cnt = *fb_count < fb_max ? *fb_count : fb_max;
    if (b >= cnt) return;
    const int64_t i = row0 + fb_rows[b];
    for (int k = threadIdx.x; k < d; k += 256) fb_feat[(int64_t)b * d + k] = feat[i * d + k];
    if (threadIdx.x == 0) fb_sqn[b] = sqn[i];
}
__global__ __launch_bounds__(256) void rr2_fb_scatter_kernel(const int *__restrict__ fb_rows, const unsigned *__restrict__ fb_count,
                                                             unsigned fb_max, const float *__restrict__ fb_rowmax,
                                                             const int *__restrict__ fb_rank, int KR,
                                                             const float *__restrict__ fb_D, int64_t ld,
                                                             float *__restrict__ rowmax, int *__restrict__ rank,
                                                             float *__restrict__ rankd) {
    const unsigned b = blockIdx.x;
    const unsigned cnt = *fb_count < fb_max ? *fb_count : fb_max;
    if (b >= cnt) return;
    const int64_t i = fb_rows[b];
    for (int t = threadIdx.x; t < KR; t += 256) {
        const int j = fb_rank[(int64_t)b * KR + t];
        rank[i * KR + t] = j;
        rankd[i * KR + t] = fb_D[(int64_t)b * ld + j];
    }
    if (threadIdx.x == 0) rowmax[i] = fb_rowmax[b];
}

// ---------------------------------------------------------------------------------------------
// host driver
// ---------------------------------------------------------------------------------------------
static uint16_t f64_to_f16_host(double d) {
    // single rounding double -> half (numpy casts the Python float 1 - lambda straight to float16)
    if (d != d) return 0x7e00u;
    uint16_t sign = 0;
    if (d < 0 || (d == 0 && 1.0 / d < 0)) {
        sign = 0x8000u;
        d = -d;
    }
    if (d >= 65520.0) return (uint16_t)(sign | 0x7c00u);
    if (d == 0.0) return sign;
    int e;
    (void)frexp(d, &e);
    const int ue = e - 1;
    const double q = (ue < -14) ? ldexp(1.0, -24) : ldexp(1.0, ue - 10);
    const double r = nearbyint(d / q) * q;
    return (uint16_t)(sign | mpreid_f32_to_f16((float)r));
}

struct StageTimer {
    bool on;
    hipStream_t s;
    std::vector<hipEvent_t> ev;
    StageTimer(bool on_, hipStream_t s_) : on(on_), s(s_) {}
    void mark() {
        if (!on) return;
        hipEvent_t e;
        (void)hipEventCreate(&e);
        (void)hipEventRecord(e, s);
        ev.push_back(e);
    }
    float ms(size_t a, size_t b) {
        float m = 0.f;
        if (on && b < ev.size()) (void)hipEventElapsedTime(&m, ev[a], ev[b]);
        return m;
    }
    ~StageTimer() {
        for (auto e : ev) (void)hipEventDestroy(e);
    }
};

template <typename F>
static int set_dyn_lds(F kernel, size_t bytes) {
    if (bytes > 160 * 1024) {
        mpreid_set_error("kernel needs %zu bytes of LDS (> 160 KiB)", bytes);
        return MPREID_ERR_UNSUPPORTED;
    }
    if (bytes > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)bytes));
    return MPREID_OK;
}

// the k-reciprocal kernel's LDS request, with an error that names the limit in the caller's terms (include/mpreid.h "Limits")
template <typename F>
static int set_krecip_lds(F kernel, size_t bytes, int K, int64_t N) {
    if (bytes > 160 * 1024) {
        mpreid_set_error("re_ranking: k1 = %d at N = %lld needs %zu bytes of LDS for the expansion lists of one row; the limit is "
                         "160 KiB per workgroup (k1 <= ~190 at N >= 20000, k1 <= 255 for N <= 18000; include/mpreid.h). The "
                         "reference (utils/reranking.py:29) takes any k1 and is called with k1 = 50",
                         K - 1, (long long)N, bytes);
        return MPREID_ERR_UNSUPPORTED;
    }
    return set_dyn_lds(kernel, bytes);
}

// max of x[0..n) -> out[0] (one workgroup)
__global__ __launch_bounds__(1024) void max_i32_kernel(const int *__restrict__ x, int64_t n, int *__restrict__ out) {
    __shared__ int part[16];
    int m = 0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) m = max(m, x[i]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = max(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) m = max(m, part[w]);
        out[0] = m;
    }
}

// inverted index of the ELL rows (fcnt, fidx, fval; row stride qcap) -> cptr / crow / cval, over the rows [nq, N) ONLY:
// the Jaccard stage produces final_dist[:nq, nq:], i.e. the accumulators of the first nq rows are never read (the
// reference computes and discards them, utils/reranking.py:84-100) -- a fifth of the entries, gathers and accumulator
// rows at nq = N / 5.  With a block-histogram buffer (chist: [CSC_B][N] u32 + the chunk-boundary table) the atomics-free
// counting sort over CSC_B blocks of the indexed rows is used (and *blocked = true: the Jaccard stage then gathers
// exact sub-ranges per row chunk); otherwise the round-1 atomic build.
static int launch_csc(int64_t N, int64_t nq, const int *fcnt, const int *fidx, const uint16_t *fval, int qcap, unsigned *ccnt,
                      unsigned *chist, long long *cptr, int *crow, uint16_t *cval, hipStream_t stream, bool *blocked) {
    static const bool csc_atomic = mpreid_tune("csc_atomic", 0) != 0;   // A/B switch: the round-1 atomic build
    const bool blocked_csc = chist && !csc_atomic && (uint64_t)N * (uint64_t)qcap < (1ull << 32);
    *blocked = blocked_csc;
    const int64_t M = N - nq;   // indexed rows
    if (blocked_csc) {
        const JaccardPlan jp = jaccard_plan(M);
        const int nranges = (int)((N + CSC_CR - 1) / CSC_CR);
        const size_t lds = (size_t)std::min<int64_t>(N, CSC_CR) * 4;
        int rc = set_dyn_lds(csc2_hist_kernel, lds);
        if (rc) return rc;
        rc = set_dyn_lds(csc2_fill_kernel, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(csc2_hist_kernel, dim3(CSC_B, nranges), dim3(1024), lds, stream, N, fcnt, fidx, qcap, jp.rpb, chist,
                           nq, (int64_t)0, N);
        hipLaunchKernelGGL(csc2_colscan_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream, N, chist, ccnt,
                           (int64_t)0, N);
        {
            const int nt = (int)((N + 1023) / 1024);
            unsigned long long *tsum = (unsigned long long *)((char *)chist + align_up((size_t)N * (CSC_B + 257) * 4, 8));   // behind the bounds table
            hipLaunchKernelGGL(scan_tile_sums_kernel, dim3((unsigned)nt), dim3(256), 0, stream, N, ccnt, tsum);
            hipLaunchKernelGGL(scan_tile_bases_kernel, dim3(1), dim3(1024), 0, stream, nt, tsum, cptr + N);
            hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)nt), dim3(256), 0, stream, N, ccnt, tsum, cptr);
        }
        hipLaunchKernelGGL(csc2_fill_kernel, dim3(CSC_B, nranges), dim3(1024), lds, stream, N, fcnt, fidx, fval, qcap, jp.rpb,
                           chist, cptr, (unsigned *)crow, nq, jp.bpc, (int64_t)0, N);   // packed entries live in the crow buffer
        hipLaunchKernelGGL(csc2_bounds_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream, N, jp.nchunks, jp.bpc,
                           chist, cptr, chist + (size_t)N * CSC_B, (int64_t)0, N);
    } else {
        HIP_TRY(hipMemsetAsync(ccnt, 0, (size_t)(N + 1) * 4, stream));
        hipLaunchKernelGGL(csc_count_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream, N, fcnt, fidx, qcap, ccnt, nq);
        hipLaunchKernelGGL(csc_scan_kernel, dim3(1), dim3(1024), 0, stream, N, ccnt, cptr);
        hipLaunchKernelGGL(csc_fill_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream, N, fcnt, fidx, fval, qcap,
                           cptr, ccnt, crow, cval, nq);
    }
    LAUNCH_CHECK();
    return MPREID_OK;
}

// Jaccard + blend for the query rows [q0, q0 + qrows): MT / rowmax / out indexed by the local row
static int launch_jaccard(int64_t N, int64_t nq, int q0, int64_t qrows, const float *MT, int64_t ld, const float *rowmax,
                          const int *fcnt, const int *fidx, const uint16_t *fval, int qcap, const long long *cptr,
                          const int *crow, const uint16_t *cval, const unsigned *chist, bool blocked, double lambda_value,
                          float *out, int64_t ldo, unsigned long long *pair_counter, hipStream_t stream,
                          const unsigned *bounds = nullptr) {   // bounds: the chunk-boundary table when it does not sit behind chist
    const uint16_t oml = f64_to_f16_host(1.0 - lambda_value);
    const float lam32 = (float)lambda_value;
    int rch = (int)std::min<int64_t>(N - nq, 49152);
    rch = (int)align_up((size_t)rch, 8);
    const unsigned *Hp = nullptr;
    int rpb = 0, bpc = 0, nchunks = 1, threads = JT;
    if (blocked) {
        const JaccardPlan jp = jaccard_plan(N - nq);
        rpb = jp.rpb; bpc = jp.bpc; nchunks = jp.nchunks; rch = jp.rch;
        threads = jp.wave_form ? 64 : JT;
        Hp = bounds ? bounds : chist + (size_t)N * CSC_B;   // the chunk-boundary table
    }
    // timing ablations (wrong results): 1 nothing is accumulated, 4 no output pass, 8 no direct path for long columns.
    // (An ablation that points every gather at ONE address measures an L2 hot spot, not the loop: removed.)
    // Compiled in only with -DMPREID_ABLATION (never in the shipped library: a stray environment variable must not be
    // able to change results).
    static const int jdbg = mpreid_ablation_env("MPREID_JACCARD_DBG");
    const size_t lds = align_up((size_t)rch * 2 + 16, 16) + (size_t)qcap * (8 + 4 + 2) + 16;
#define MPREID_JACCARD_LAUNCH(JT_, NPF_, PD_, PK_)                                                                       \
    {                                                                                                                    \
        int rc = set_dyn_lds(jaccard_kernel<JT_, NPF_, PD_, PK_>, lds);                                                  \
        if (rc) return rc;                                                                                               \
        hipLaunchKernelGGL((jaccard_kernel<JT_, NPF_, PD_, PK_>), dim3((unsigned)qrows, (unsigned)nchunks), dim3(JT_),   \
                           lds, stream, N, nq, MT, ld, rowmax, fcnt, fidx, fval, qcap, cptr, crow, cval, rch, oml, lam32, \
                           out, ldo, pair_counter, q0, Hp, rpb, bpc, jdbg);                                              \
    }
    // (multi-wave form, N = 20 000 / Market shape: 512 threads <2, 4> 1.58 / 1.42 ms, 256 threads <3, 4> 1.36 / 1.21,
    // <3, 8> the same, <2, 4> 2.05 (columns longer than 512 entries take the direct path), 128 threads <5, 4> 1.72 / 1.50)
    // blocked index = packed entries (csc2_fill_kernel); the atomic build keeps (row, value) in two arrays
    static const int jtab = mpreid_tune("jaccard_table", 0);   // A/B: LDS table form
    const size_t wl = align_up((size_t)rch * 2 + 16, 16);   // table-free kernels: the accumulators only
    if (threads == 64 && !jtab) {
        int rc = set_dyn_lds(jaccard_wave_kernel<64, 2, 8>, wl);
        if (rc) return rc;
        hipLaunchKernelGGL((jaccard_wave_kernel<64, 2, 8>), dim3((unsigned)qrows, (unsigned)nchunks), dim3(64), wl, stream, N, nq,
                           MT, ld, rowmax, fcnt, fidx, fval, qcap, cptr, (const unsigned *)crow, rch, oml, lam32, out, ldo,
                           pair_counter, q0, Hp, rpb, bpc);
    } else if (threads == 64) MPREID_JACCARD_LAUNCH(64, 2, 8, true)
    else if (blocked && !jtab) {
        int rc = set_dyn_lds(jaccard_wave_kernel<JT, 3, 4>, wl);
        if (rc) return rc;
        hipLaunchKernelGGL((jaccard_wave_kernel<JT, 3, 4>), dim3((unsigned)qrows, (unsigned)nchunks), dim3(JT), wl, stream, N, nq,
                           MT, ld, rowmax, fcnt, fidx, fval, qcap, cptr, (const unsigned *)crow, rch, oml, lam32, out, ldo,
                           pair_counter, q0, Hp, rpb, bpc);
    }
    else if (blocked) MPREID_JACCARD_LAUNCH(JT, 3, 4, true)
    else MPREID_JACCARD_LAUNCH(JT, 3, 4, false)
#undef MPREID_JACCARD_LAUNCH
    LAUNCH_CHECK();
    return MPREID_OK;
}

// Everything after the V rows (query expansion, inverted index, Jaccard + blend, statistics): shared by the dense and
// the sparse algorithm.  MT = distance rows of (at least) the queries, row stride ld; rowmax indexed like MT's rows.
// The ONE host round trip of the call is in here: 16 bytes (largest union size of the query expansion, which sizes
// the ELL rows of V_qe and the LDS of the kernels that walk them, and nnz(V)); counts and sums are reduced on the GPU.
struct TailArgs {
    int64_t N, nq;
    int k1, k2, KR, h, vcap;
    int64_t qcap_bound;   // capacity the workspace was sized for (entries per row of V_qe / of the inverted index)
    const int *rank;
    int *vcnt, *vidx;
    uint16_t *vval;
    int *ucnt, *qcnt, *qidx;
    uint16_t *qval;
    unsigned *ccnt;
    unsigned *chist;                // [CSC_B][N] block histograms of the atomics-free inverted-index build (or NULL)
    long long *cptr;
    int *crow;
    uint16_t *cval;
    unsigned long long *counters;   // [0] Jaccard pairs, [1] sum |R|, [2] nnz(V), [3] max union (int), [4] fallback rows, [5] candidates, [6] nnz(V_qe)
    const float *MT;
    int64_t ld;
    const float *rowmax;
    float *out;
    int64_t ldo;
    double lambda_value;
    int algo;
    hipEvent_t join;   // (sparse) the distance rows MT are produced on a side stream: the Jaccard stage waits for this
    // (sparse, no side stream) launches the exact distance rows on the main stream; the tail calls it BETWEEN the
    // device-side sizing of the V_qe rows and the host's wait for that size, so that the host round trip happens while
    // the GPU works on the rows instead of leaving a ~40 us bubble
    std::function<int(hipStream_t)> mid_launch;
    hipEvent_t sized;   // recorded after the size has been copied to the host buffer

};

static int rerank_tail(const TailArgs &a, hipStream_t stream, StageTimer &tm, mpreid_rerank_stats *stats, int marks_so_far) {
    const int64_t N = a.N;
    const int nw = (int)((N + 31) >> 5);
    const int k2e = (int)std::min<int64_t>(a.k2, N); // rows the query expansion really averages (numpy clamps the slice)
    int qcap = a.vcap;
    const int *fcnt = a.vcnt, *fidx = a.vidx;
    const uint16_t *fval = a.vval;
    bool mid_done = false;
    // (7) query expansion
    hipLaunchKernelGGL(sum_i32_kernel, dim3(1), dim3(1024), 0, stream, a.vcnt, N, a.counters + 2);
    if (a.k2 != 1) {
        {
            const size_t lds = (size_t)nw * 4;
            int rc = set_dyn_lds(qe_count_kernel, lds);
            if (rc) return rc;
            hipLaunchKernelGGL(qe_count_kernel, dim3((unsigned)N), dim3(64), lds, stream, N, a.rank, a.KR, k2e, a.vcnt, a.vidx,
                               a.vcap, a.ucnt, 0);
            hipLaunchKernelGGL(max_i32_kernel, dim3(1), dim3(1024), 0, stream, a.ucnt, N, (int *)(a.counters + 3));
            LAUNCH_CHECK();
        }
        int mxu = 1;
        HIP_TRY(hipMemcpyAsync(&mxu, a.counters + 3, 4, hipMemcpyDeviceToHost, stream));
        tm.mark(); // +1
        if (a.mid_launch && a.sized) {
            HIP_TRY(hipEventRecord(a.sized, stream));
            int rc = a.mid_launch(stream);
            if (rc) return rc;
            tm.mark(); // +2
            HIP_TRY(hipEventSynchronize(a.sized));
        } else {
            tm.mark(); // +2
            HIP_TRY(hipStreamSynchronize(stream));
        }
        mid_done = true;
        if (mxu < 1) mxu = 1;
        qcap = (int)align_up((size_t)mxu, 8);
        if ((int64_t)qcap > a.qcap_bound) {
            if ((int64_t)mxu <= a.qcap_bound) {
                qcap = (int)a.qcap_bound;
            } else {
                mpreid_set_error("query expansion: a row has %d entries, the workspace was sized for %lld; retry with "
                                 "MPREID_RERANK_DENSE", mxu, (long long)a.qcap_bound);
                return MPREID_ERR_RETRY_DENSE;
            }
        }
        {
            const size_t lds = (size_t)nw * 8 + (size_t)qcap * 8;
            int rc = set_dyn_lds(qe_fill_kernel, lds);
            if (rc) return rc;
            hipLaunchKernelGGL(qe_fill_kernel, dim3((unsigned)N), dim3(64), lds, stream, N, a.rank, a.KR, k2e, a.vcnt, a.vidx,
                               a.vval, a.vcap, qcap, a.qcnt, a.qidx, a.qval, 0);
            LAUNCH_CHECK();
        }
        fcnt = a.qcnt;
        fidx = a.qidx;
        fval = a.qval;
    }
    if (!mid_done) {   // k2 == 1: no expansion, no sizing
        tm.mark(); // +1
        if (a.mid_launch) {
            int rc = a.mid_launch(stream);
            if (rc) return rc;
        }
        tm.mark(); // +2
    }
    hipLaunchKernelGGL(sum_i32_kernel, dim3(1), dim3(1024), 0, stream, fcnt, N, a.counters + 6);   // nnz(V_qe), all rows
    tm.mark(); // +3
    // inverted index
    bool blocked_csc = false;
    {
        int rc = launch_csc(N, a.nq, fcnt, fidx, fval, qcap, a.ccnt, a.chist, a.cptr, a.crow, a.cval, stream, &blocked_csc);
        if (rc) return rc;
    }
    tm.mark(); // +4
    // (8)-(11) Jaccard + blend
    if (a.join) HIP_TRY(hipStreamWaitEvent(stream, a.join, 0));
    {
        int rc = launch_jaccard(N, a.nq, 0, a.nq, a.MT, a.ld, a.rowmax, fcnt, fidx, fval, qcap, a.cptr, a.crow, a.cval, a.chist,
                                blocked_csc, a.lambda_value, a.out, a.ldo, a.counters, stream);
        if (rc) return rc;
    }
    tm.mark(); // +5
    unsigned long long cnt[7] = {0, 0, 0, 0, 0, 0, 0};
    long long nnz_total = 0;
    HIP_TRY(hipMemcpyAsync(cnt, a.counters, sizeof(cnt), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(&nnz_total, a.cptr + N, 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (stats) {
        const int m = marks_so_far;   // index of the mark taken after the V rows
        stats->n = N;
        stats->k1 = a.k1;
        stats->k2 = a.k2;
        stats->half_k1 = a.h;
        stats->v_cap = a.vcap;
        stats->vqe_cap = qcap;
        stats->v_nnz = (int64_t)cnt[2];
        stats->vqe_nnz = (int64_t)cnt[6];   // (nnz_total = entries of the inverted index: gallery rows only)
        stats->jaccard_pairs = (int64_t)cnt[0];
        stats->krecip_r_sum = (int64_t)cnt[1];
        stats->fallback_rows = (int64_t)(cnt[4] & 0xffffffffull);
        stats->cand_total = (int64_t)cnt[5];
        stats->algo = a.algo;
        stats->ms_gemm = tm.ms(0, 1);
        stats->ms_topk = tm.ms(1, 2);
        stats->ms_krecip = tm.ms(2, 3);
        stats->ms_dq = a.mid_launch ? tm.ms(m + 1, m + 2) : 0.0f;   // (side stream: filled in by the caller)
        stats->ms_qe = tm.ms(m, m + 1) + tm.ms(m + 2, m + 3);
        stats->ms_csc = tm.ms(m + 3, m + 4);
        stats->ms_jaccard = tm.ms(m + 4, m + 5);
        stats->ms_total = tm.ms(0, m + 5);
    }
    return MPREID_OK;
}

static int rerank_dense(const float *q, const float *g, int64_t nq, int64_t ng, int d, int k1, int k2,
                                 double lambda_value, const float *local, int only_local, float *out, int64_t ldo,
                                 void *ws, size_t ws_bytes, mpreid_stream_t stream_, mpreid_rerank_stats *stats,
                                 int timing) {
    ARG_CHECK(q && g && out && nq > 0 && ng > 0 && d > 0 && k1 >= 0 && k2 >= 1 && ldo >= ng);
    ARG_CHECK(!only_local || local);
    const RerankLayout L = make_layout(nq, ng, d, k1, k2, local != nullptr);
    const int64_t N = L.N;
    ARG_CHECK(L.KR <= N);
    if (L.KR > 256) {
        mpreid_set_error("re_ranking: max(k1 + 1, k2) = %d exceeds this build's limit of 256 (the neighbour selection sorts its "
                         "winners in one 256-entry LDS network and the reciprocity masks hold 256 bits per row; include/mpreid.h). "
                         "The reference (utils/reranking.py:29) takes any k and is called with k1 = 50, k2 = 15", L.KR);
        return MPREID_ERR_UNSUPPORTED;
    }
    if (N >= (1ll << 31) - 64) {
        mpreid_set_error("N too large");
        return MPREID_ERR_UNSUPPORTED;
    }
    if (!ws || ws_bytes < L.total) {
        mpreid_set_error("rerank workspace too small: %zu < %zu", ws_bytes, L.total);
        return MPREID_ERR_WORKSPACE;
    }
    hipStream_t stream = (hipStream_t)stream_;
    char *base = (char *)ws;
    float *feat = (float *)(base + L.feat), *norms = (float *)(base + L.norms);
    float *D = (float *)(base + L.D), *MT = (float *)(base + L.MT), *rowmax = (float *)(base + L.rowmax);
    int *rank = (int *)(base + L.rank), *vcnt = (int *)(base + L.vcnt), *vidx = (int *)(base + L.vidx);
    uint16_t *vval = (uint16_t *)(base + L.vval);
    int *ucnt = (int *)(base + L.ucnt), *qcnt = (int *)(base + L.qcnt), *qidx = (int *)(base + L.qidx);
    uint16_t *qval = (uint16_t *)(base + L.qval);
    unsigned *ccnt = (unsigned *)(base + L.ccnt);
    long long *cptr = (long long *)(base + L.cptr);
    int *crow = (int *)(base + L.crow);
    uint16_t *cval = (uint16_t *)(base + L.cval);
    unsigned long long *counters = (unsigned long long *)(base + L.counters);
    const int nw = (int)((N + 31) >> 5);

    StageTimer tm(timing != 0, stream);
    HIP_TRY(hipMemsetAsync(counters, 0, 64, stream));
    tm.mark(); // 0
    // (1) original_dist
    if (!only_local) {
        HIP_TRY(hipMemcpyAsync(feat, q, (size_t)nq * d * 4, hipMemcpyDeviceToDevice, stream));
        HIP_TRY(hipMemcpyAsync(feat + (size_t)nq * d, g, (size_t)ng * d * 4, hipMemcpyDeviceToDevice, stream));
        int rc = mpreid_sqnorm_f32(feat, N, d, norms, stream);
        if (rc) return rc;
        rc = mpreid_distance_launch(feat, feat, N, N, d, norms, norms, D, L.ld, 0, stream);
        if (rc) return rc;
    }
    if (local) {
        const dim3 grid((unsigned)((N + 31) / 32), (unsigned)((N + 31) / 32));
        hipLaunchKernelGGL(make_mt_kernel, grid, dim3(256), 0, stream, only_local ? (const float *)nullptr : D, L.ld,
                           local, N, MT, L.ld);
        LAUNCH_CHECK();
    }
    tm.mark(); // 1
    // (2)+(3) row max and the first KR neighbours in (value, index) order
    {
        int rc = launch_rowmax_topk(MT, L.ld, N, L.KR, N, rowmax, rank, stream);
        if (rc) return rc;
    }
    tm.mark(); // 2
    // (4)-(6) V rows
    {
        const size_t lds = krecip_lds_bytes(false, nw, L.K, L.vcap, 0);
        int rc = set_krecip_lds(krecip_kernel<false>, lds, L.K, N);
        if (rc) return rc;
        unsigned *rk = (unsigned *)(base + L.rbits);
        hipLaunchKernelGGL(recip_bits_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, stream, rank, N, L.K, L.KR, L.h, rk,
                           rk + (size_t)N * RB_WORDS);
        hipLaunchKernelGGL(krecip_kernel<false>, dim3((unsigned)N), dim3(256), lds, stream, MT, L.ld, N, rowmax, rank, L.K,
                           L.KR, L.h, L.vcap, vcnt, vidx, vval, 0, ucnt, (const float *)nullptr,
                           (const float *)nullptr, 0, (const float *)nullptr, rk, rk + (size_t)N * RB_WORDS);
        hipLaunchKernelGGL(sum_i32_kernel, dim3(1), dim3(1024), 0, stream, ucnt, N, counters + 1);
        LAUNCH_CHECK();
    }
    tm.mark(); // 3
    TailArgs ta{};
    ta.N = N; ta.nq = nq; ta.k1 = k1; ta.k2 = k2; ta.KR = L.KR; ta.h = L.h; ta.vcap = L.vcap; ta.qcap_bound = L.qcap_bound;
    ta.rank = rank; ta.vcnt = vcnt; ta.vidx = vidx; ta.vval = vval; ta.ucnt = ucnt; ta.qcnt = qcnt; ta.qidx = qidx;
    ta.qval = qval; ta.ccnt = ccnt; ta.cptr = cptr; ta.crow = crow; ta.cval = cval; ta.counters = counters;
    ta.chist = (unsigned *)(base + L.chist);
    ta.MT = MT; ta.ld = L.ld; ta.rowmax = rowmax; ta.out = out; ta.ldo = ldo; ta.lambda_value = lambda_value;
    ta.algo = MPREID_RERANK_DENSE;
    return rerank_tail(ta, stream, tm, stats, 3);
}

// ---------------------------------------------------------------------------------------------
// sparse algorithm: host driver (kernels: the "Candidate pipeline" section above)
// ---------------------------------------------------------------------------------------------
constexpr int RR2_CAP_LO = 384, RR2_CAP_HI = 128, RR2_SAMPLE_STRIDE = 16, RR2_RSEL = 10, RR2_QCAP = 4096;

struct Rerank2Layout {
    int64_t N, Np, Nsp, ld, fb_max;
    int dp, K, KR, h, vcap;
    int64_t qcap_bound;
    size_t feat, sqn, feat16, samp16, sampn, sampD, tlo, thi, eps, cnt_lo, cnt_hi, list_lo, list_hi, gstat, rowmax, rank, rankd, rbits,
        fb_count, fb_rows, fb_feat, fb_sqn, fb_D, fb_rowmax, fb_rank, vcnt, vidx, vval, ucnt, qcnt, qidx, qval, ccnt, chist,
        cptr, crow, cval, dq, counters, dq_ws, dq_ws_bytes, total;
};

static Rerank2Layout make_layout2(int64_t nq, int64_t ng, int d, int k1, int k2, bool split3_rows = false) {
    Rerank2Layout L{};
    L.N = nq + ng;
    L.Np = (int64_t)align_up((size_t)L.N, 256);
    const int64_t ns = (L.N + RR2_SAMPLE_STRIDE - 1) / RR2_SAMPLE_STRIDE;
    L.Nsp = (int64_t)align_up((size_t)ns, 256);
    L.ld = (int64_t)align_up((size_t)L.N, 64);
    L.dp = (int)align_up((size_t)d, 64);
    L.fb_max = std::max<int64_t>(256, L.N / 16);
    L.K = (int)std::min<int64_t>(k1 + 1, L.N);
    L.KR = std::max(L.K, (int)std::min<int64_t>(k2, L.N));
    L.h = (int)std::min<int64_t>(mpreid_half_k1(k1), L.N);
    L.vcap = (int)std::min<int64_t>((int64_t)L.K * (1 + L.h), L.N);
    L.qcap_bound = std::min<int64_t>(std::min<int64_t>(L.N, (int64_t)std::max(k2, 1) * L.vcap), RR2_QCAP);
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += align_up(bytes, 256);
        return o;
    };
    const size_t N = (size_t)L.N, Np = (size_t)L.Np, F = (size_t)L.fb_max;
    L.feat = take(N * (size_t)d * 4);
    L.sqn = take(Np * 4);
    L.feat16 = take(Np * (size_t)L.dp * 2);
    L.samp16 = take((size_t)L.Nsp * L.dp * 2);
    L.sampn = take((size_t)L.Nsp * 4);
    L.sampD = take(Np * (size_t)L.Nsp * 4);
    L.tlo = take(Np * 4);
    L.thi = take(Np * 4);
    L.eps = take(Np * 4);
    L.cnt_lo = take(Np * 4);
    L.cnt_hi = take(Np * 4);
    L.list_lo = take(Np * (size_t)RR2_CAP_LO * 8);
    L.list_hi = take(Np * (size_t)RR2_CAP_HI * 8);
    L.gstat = take(64);
    L.rowmax = take(N * 4);
    L.rank = take(N * (size_t)L.KR * 4);
    L.rankd = take(N * (size_t)L.KR * 4);
    L.rbits = take(N * (size_t)8 * 4 * 2);
    L.fb_count = take(64);
    L.fb_rows = take(F * 4);
    L.fb_feat = take(F * (size_t)d * 4);
    L.fb_sqn = take(F * 4);
    L.fb_D = take(F * (size_t)L.ld * 4);
    L.fb_rowmax = take(F * 4);
    L.fb_rank = take(F * (size_t)L.KR * 4);
    L.vcnt = take(N * 4);
    L.vidx = take(N * (size_t)L.vcap * 4);
    L.vval = take(N * (size_t)L.vcap * 2);
    L.ucnt = take(N * 4);
    L.qcnt = take(N * 4);
    L.qidx = take(N * (size_t)L.qcap_bound * 4);
    L.qval = take(N * (size_t)L.qcap_bound * 2);
    L.ccnt = take((N + 1) * 4);
    L.chist = take(csc_hist_bytes((int64_t)N));   // CSC_B block histograms + chunk bounds
    L.cptr = take((N + 1) * 8);
    L.crow = take(N * (size_t)L.qcap_bound * 4);
    L.cval = take(N * (size_t)L.qcap_bound * 2);
    L.dq = take((size_t)nq * L.ld * 4);
    L.counters = take(64);
    L.dq_ws_bytes = split3_rows ? mpreid_distance_split3_ws_bytes(nq, ng, d) : 0;
    L.dq_ws = take(L.dq_ws_bytes);
    L.total = off;
    return L;
}

// the sparse algorithm applies when there is no local_distmat, the problem is large enough for the sample to mean
// something, and the neighbour count fits the refinement kernel's evaluation budget
static bool sparse_eligible(int64_t nq, int64_t ng, int k1, int k2, const float *local) {
    const int64_t N = nq + ng;
    const int64_t KR = std::max<int64_t>(std::min<int64_t>(k1 + 1, N), std::min<int64_t>(k2, N));
    return local == nullptr && N >= 2048 && KR <= 64;
}

// A second HIP stream (per device, created once) for work that is independent of the main chain: the exact distance
// rows of the queries depend only on the features, so the fp32 matrix pipe computes them while the candidate
// refinement / expansion / inverted-index kernels (gathers, LDS, no MFMA) run on the caller's stream.  Fork and join
// with events; to the caller the call is still ordered on `stream`.
struct SideStream {
    hipStream_t s = nullptr;
    hipEvent_t fork = nullptr, join = nullptr, t0 = nullptr, t1 = nullptr;
    int dev = 0;
};
// Side streams are LEASED per call from a per-device pool (created on demand, returned at the end of the call), so that
// calls on different streams / devices of one process never share one and need no process-wide lock.  The lease's
// destructor runs on EVERY exit path: when work was forked onto the side stream it waits for that work first, so the
// workspace (the distance rows it writes) is quiescent when the call returns -- also on the early MPREID_ERR_RETRY_DENSE
// / error returns, after which the caller typically frees or re-purposes the workspace.
struct SideLease {
    SideStream *ss = nullptr;
    bool forked = false;
    static std::mutex &mu() {
        static std::mutex m;
        return m;
    }
    static std::vector<SideStream *> &pool() {
        static std::vector<SideStream *> p;
        return p;
    }
    int acquire() {
        if (ss) return MPREID_OK;
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        {
            std::lock_guard<std::mutex> lk(mu());
            auto &p = pool();
            for (size_t i = 0; i < p.size(); ++i)
                if (p[i]->dev == dev) {
                    ss = p[i];
                    p.erase(p.begin() + (long)i);
                    return MPREID_OK;
                }
        }
        SideStream *n = new SideStream();
        n->dev = dev;
        if (hipStreamCreateWithFlags(&n->s, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&n->fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&n->join, hipEventDisableTiming) != hipSuccess || hipEventCreate(&n->t0) != hipSuccess ||
            hipEventCreate(&n->t1) != hipSuccess) {
            mpreid_set_error("re-ranking: could not create the side stream");
            delete n;   // (handles created so far leak on this never-seen path)
            return MPREID_ERR_NODEVICE;
        }
        ss = n;
        return MPREID_OK;
    }
    ~SideLease() {
        if (!ss) return;
        if (forked) (void)hipStreamSynchronize(ss->s);
        std::lock_guard<std::mutex> lk(mu());
        pool().push_back(ss);
    }
};

static int rerank_sparse(const float *q, const float *g, int64_t nq, int64_t ng, int d, int k1, int k2, double lambda_value,
                         float *out, int64_t ldo, void *ws, size_t ws_bytes, mpreid_stream_t stream_,
                         mpreid_rerank_stats *stats, int timing, bool split3_rows) {
    ARG_CHECK(q && g && out && nq > 0 && ng > 0 && d > 0 && k1 >= 0 && k2 >= 1 && ldo >= ng);
    const Rerank2Layout L = make_layout2(nq, ng, d, k1, k2, split3_rows);
    const int64_t N = L.N;
    if (N >= (1ll << 28) - 512) {
        mpreid_set_error("N too large for the sparse algorithm");
        return MPREID_ERR_UNSUPPORTED;
    }
    if (!ws || ws_bytes < L.total) {
        mpreid_set_error("rerank workspace too small: %zu < %zu", ws_bytes, L.total);
        return MPREID_ERR_WORKSPACE;
    }
    hipStream_t stream = (hipStream_t)stream_;
    SideLease lease;   // (declared before anything that can return: its destructor joins the side stream on every path)
    char *base = (char *)ws;
    float *feat = (float *)(base + L.feat), *sqn = (float *)(base + L.sqn);
    _Float16 *feat16 = (_Float16 *)(base + L.feat16), *samp16 = (_Float16 *)(base + L.samp16);
    float *sampn = (float *)(base + L.sampn), *sampD = (float *)(base + L.sampD);
    float *tlo = (float *)(base + L.tlo), *thi = (float *)(base + L.thi), *eps = (float *)(base + L.eps);
    unsigned *cnt_lo = (unsigned *)(base + L.cnt_lo), *cnt_hi = (unsigned *)(base + L.cnt_hi);
    uint2 *list_lo = (uint2 *)(base + L.list_lo), *list_hi = (uint2 *)(base + L.list_hi);
    float *gstat = (float *)(base + L.gstat), *rowmax = (float *)(base + L.rowmax);
    int *rank = (int *)(base + L.rank);
    float *rankd = (float *)(base + L.rankd);
    unsigned *fb_count = (unsigned *)(base + L.fb_count);
    int *fb_rows = (int *)(base + L.fb_rows);
    float *fb_feat = (float *)(base + L.fb_feat), *fb_sqn = (float *)(base + L.fb_sqn), *fb_D = (float *)(base + L.fb_D),
          *fb_rowmax = (float *)(base + L.fb_rowmax);
    int *fb_rank = (int *)(base + L.fb_rank);
    int *vcnt = (int *)(base + L.vcnt), *vidx = (int *)(base + L.vidx);
    uint16_t *vval = (uint16_t *)(base + L.vval);
    int *ucnt = (int *)(base + L.ucnt);
    float *dq = (float *)(base + L.dq);
    unsigned long long *counters = (unsigned long long *)(base + L.counters);
    const int nw = (int)((N + 31) >> 5);
    const int ns = (int)((N + RR2_SAMPLE_STRIDE - 1) / RR2_SAMPLE_STRIDE);

    StageTimer tm(timing != 0, stream);
    HIP_TRY(hipMemsetAsync(counters, 0, 64, stream));
    HIP_TRY(hipMemsetAsync(fb_count, 0, 64, stream));
    // rows that need the fallback but do not fit its buffer keep these values: the call then ends with
    // MPREID_ERR_RETRY_DENSE, but every kernel on the way must still see valid neighbour indices
    HIP_TRY(hipMemsetAsync(rank, 0, (size_t)N * L.KR * 4, stream));
    HIP_TRY(hipMemsetAsync(rankd, 0, (size_t)N * L.KR * 4, stream));
    HIP_TRY(hipMemsetAsync(rowmax, 0, (size_t)N * 4, stream));
    tm.mark(); // 0
    // features, exact squared norms (padded with zeros), fp16 operands, sample
    HIP_TRY(hipMemcpyAsync(feat, q, (size_t)nq * d * 4, hipMemcpyDeviceToDevice, stream));
    HIP_TRY(hipMemcpyAsync(feat + (size_t)nq * d, g, (size_t)ng * d * 4, hipMemcpyDeviceToDevice, stream));
    HIP_TRY(hipMemsetAsync(sqn, 0, (size_t)L.Np * 4, stream));
    int rc = mpreid_sqnorm_f32(feat, N, d, sqn, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(rr2_norm_stats_kernel, dim3(1), dim3(1024), 0, stream, sqn, N, gstat);
    hipLaunchKernelGGL(rr2_cast_rows_kernel, dim3((unsigned)L.Np), dim3(256), 0, stream, feat, sqn, N, d, 1, feat16, L.dp,
                       (float *)nullptr);
    hipLaunchKernelGGL(rr2_cast_rows_kernel, dim3((unsigned)L.Nsp), dim3(256), 0, stream, feat, sqn, N, d,
                       RR2_SAMPLE_STRIDE, samp16, L.dp, sampn);
    LAUNCH_CHECK();
    {   // sample pass: [Np] x [Nsp] one-pass fp16 distances, stored
        GemmArgs a{};
        a.A = feat16; a.W = samp16; a.M = (int)L.Np; a.N = (int)L.Nsp; a.K = L.dp;
        a.out = sampD; a.ldo = L.Nsp; a.aux = sqn; a.aux2 = sampn; a.m_valid = (int)N; a.n_valid = ns;
        rc = launch_gemm_f16(a, GE_EUCLID, stream);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(rr2_threshold_kernel, dim3((unsigned)L.Np), dim3(256), 0, stream, sampD, L.Nsp, ns, N, RR2_RSEL, sqn,
                       gstat, d, tlo, thi, eps, cnt_lo, cnt_hi);
    LAUNCH_CHECK();
    {   // fused pass: symmetric N x N on the matrix cores, candidates only
        GemmArgs a{};
        a.A = feat16; a.W = feat16; a.M = (int)L.Np; a.N = (int)L.Np; a.K = L.dp;
        a.aux = sqn; a.aux2 = sqn; a.m_valid = (int)N; a.n_valid = (int)N;
        a.tlo = tlo; a.thi = thi; a.cnt_lo = cnt_lo; a.cnt_hi = cnt_hi; a.list_lo = list_lo; a.list_hi = list_hi;
        a.cap_lo = RR2_CAP_LO; a.cap_hi = RR2_CAP_HI; a.sym = 1;
        static const int cand_dbg = mpreid_ablation_env("MPREID_CAND_DBG");   // timing experiments (wrong results)
        if (cand_dbg & 1) a.sym = 0;
        if (cand_dbg & 2) a.stagger = 2;
        rc = launch_gemm_f16(a, GE_CAND, stream);
        if (rc) return rc;
    }
    tm.mark(); // 1
    // exact distance rows of the queries (what the Jaccard blend reads): only the GALLERY columns [nq, N) of the query rows are ever read (the blend of final_dist[:nq, nq:]): the first
    // nq columns of dq stay unwritten (20 % of the fp32 matrix work at nq = N / 5)
    auto launch_dq = [&](hipStream_t s, int64_t q0, int64_t rows) -> int {
        if (N == nq || rows <= 0) return MPREID_OK;
        // MPREID_RERANK_SPARSE_SPLIT3: the rows that only feed the blend term lambda * d / max come from the fp16 matrix
        // cores (3-term split, |error| <= 1e-6: |delta final| <= lambda * 1e-6 / max); everything discrete -- neighbours,
        // V, V_qe, the Jaccard term -- is computed exactly as in the bit-parity mode
        if (split3_rows)
            return mpreid_distance_f16_split3(feat + (size_t)q0 * d, feat + (size_t)nq * d, rows, N - nq, d, sqn + q0, sqn + nq,
                                              dq + (size_t)q0 * L.ld + nq, L.ld, 0, base + L.dq_ws, L.dq_ws_bytes, s);
        return mpreid_distance_launch(feat + (size_t)q0 * d, feat + (size_t)nq * d, rows, N - nq, d, sqn + q0, sqn + nq,
                                      dq + (size_t)q0 * L.ld + nq, L.ld, 0, s);
    };
    // Overlap of the fp32 matrix work with the rest (MPREID_TUNE rerank_overlap = 0 / 1 forces it off / on): one launch on
    // the side stream, forked AFTER the fused GEMM (a persistent kernel that owns every CU's LDS: forked before it, the
    // two GEMMs only take turns).  Nearly zero-sum wherever it was measured: N = 20 000 6.06 vs 6.04 ms, N = 100 000
    // 87.7 -> 86.2 ms, MSMT17 shape 77.7 -> 76.7 ms (on by default only from N = 50 000).  Also measured at N = 100 000
    // and NOT kept: the rows in four query blocks forked after the V rows so that the Jaccard stage of block b runs
    // beside the rows of block b + 1 -- 87.0 ms as launched (the GEMM's workgroups take every slot that frees up and
    // the partner starves), 90.2 ms against 90.7 in line with the GEMM made persistent at two workgroups per CU so that
    // both kernels are resident (both then run at half speed: they share each CU's load path and LDS).
    static const int ov_tune = mpreid_tune("rerank_overlap", -1);
    const int ov_mode = ov_tune >= 0 ? (ov_tune != 0) : (N < 50000 ? 0 : 1);
    const bool no_overlap = ov_mode == 0;
    SideStream *ss = nullptr;
    if (ov_mode != 0) {
        if ((rc = lease.acquire())) return rc;
        ss = lease.ss;
    }
    if (ov_mode == 1) {
        HIP_TRY(hipEventRecord(ss->fork, stream));
        HIP_TRY(hipStreamWaitEvent(ss->s, ss->fork, 0));
        lease.forked = true;
        if (timing) HIP_TRY(hipEventRecord(ss->t0, ss->s));
        if ((rc = launch_dq(ss->s, 0, nq))) return rc;
        if (timing) HIP_TRY(hipEventRecord(ss->t1, ss->s));
        HIP_TRY(hipEventRecord(ss->join, ss->s));
    }
    {   // refinement + fallback rows
        const size_t lds = 512 * 8 + RR2_MAXE * 8 + RR2_MAXH * 8 + (size_t)((d + 3) & ~3) * 4 + 2 * 64 * WXD_STRIDE * 4;
        rc = set_dyn_lds(rr2_refine_kernel<RR2_CAP_LO, RR2_CAP_HI>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL((rr2_refine_kernel<RR2_CAP_LO, RR2_CAP_HI>), dim3((unsigned)N), dim3(256), lds, stream, feat, sqn, N,
                           d, L.KR, cnt_lo, list_lo, cnt_hi, list_hi, tlo, eps, rowmax, rank, rankd, fb_count, fb_rows,
                           (unsigned)L.fb_max, (int64_t)0);
        hipLaunchKernelGGL(rr2_fb_gather_kernel, dim3((unsigned)L.fb_max), dim3(256), 0, stream, feat, sqn, d, fb_rows,
                           fb_count, (unsigned)L.fb_max, fb_feat, fb_sqn, (int64_t)0);
        LAUNCH_CHECK();
        rc = mpreid_distance_launch(fb_feat, feat, L.fb_max, N, d, fb_sqn, sqn, fb_D, L.ld, 0, stream, fb_count);
        if (rc) return rc;
        rc = launch_rowmax_topk(fb_D, L.ld, N, L.KR, L.fb_max, fb_rowmax, fb_rank, stream, fb_count);
        if (rc) return rc;
        hipLaunchKernelGGL(rr2_fb_scatter_kernel, dim3((unsigned)L.fb_max), dim3(256), 0, stream, fb_rows, fb_count,
                           (unsigned)L.fb_max, fb_rowmax, fb_rank, L.KR, fb_D, L.ld, rowmax, rank, rankd);
        hipLaunchKernelGGL(sum_i32_kernel, dim3(1), dim3(1024), 0, stream, (const int *)cnt_lo, N, counters + 5);
        HIP_TRY(hipMemcpyAsync(counters + 4, fb_count, 4, hipMemcpyDeviceToDevice, stream));
        LAUNCH_CHECK();
    }
    tm.mark(); // 2
    {   // V rows, distances on the fly
        const size_t lds = krecip_lds_bytes(true, nw, L.K, L.vcap, d);
        rc = set_krecip_lds(krecip_kernel<true>, lds, L.K, N);
        if (rc) return rc;
        unsigned *rk = (unsigned *)(base + L.rbits);
        hipLaunchKernelGGL(recip_bits_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, stream, rank, N, L.K, L.KR, L.h, rk,
                           rk + (size_t)N * RB_WORDS);
        hipLaunchKernelGGL(krecip_kernel<true>, dim3((unsigned)N), dim3(256), lds, stream, (const float *)nullptr, L.ld, N,
                           rowmax, rank, L.K, L.KR, L.h, L.vcap, vcnt, vidx, vval, 0, ucnt, feat, sqn, d, rankd, rk,
                           rk + (size_t)N * RB_WORDS);
        hipLaunchKernelGGL(sum_i32_kernel, dim3(1), dim3(1024), 0, stream, ucnt, N, counters + 1);
        LAUNCH_CHECK();
    }
    tm.mark(); // 3
    TailArgs ta{};
    SideStream *evs = nullptr;
    if (no_overlap) {   // exact distance rows of the queries, in line: launched by the tail (see TailArgs::mid_launch)
        if ((rc = lease.acquire())) return rc;     // (only its `fork` event is used here)
        evs = lease.ss;
        ta.mid_launch = [&](hipStream_t s) -> int { return launch_dq(s, 0, nq); };
        ta.sized = evs->fork;
    }
    ta.N = N; ta.nq = nq; ta.k1 = k1; ta.k2 = k2; ta.KR = L.KR; ta.h = L.h; ta.vcap = L.vcap; ta.qcap_bound = L.qcap_bound;
    ta.rank = rank; ta.vcnt = vcnt; ta.vidx = vidx; ta.vval = vval; ta.ucnt = ucnt; ta.qcnt = (int *)(base + L.qcnt);
    ta.qidx = (int *)(base + L.qidx); ta.qval = (uint16_t *)(base + L.qval); ta.ccnt = (unsigned *)(base + L.ccnt);
    ta.cptr = (long long *)(base + L.cptr); ta.crow = (int *)(base + L.crow); ta.cval = (uint16_t *)(base + L.cval);
    ta.chist = (unsigned *)(base + L.chist);
    ta.counters = counters; ta.MT = dq; ta.ld = L.ld; ta.rowmax = rowmax; ta.out = out; ta.ldo = ldo;
    ta.lambda_value = lambda_value; ta.algo = split3_rows ? MPREID_RERANK_SPARSE_SPLIT3 : MPREID_RERANK_SPARSE;
    ta.join = ss ? ss->join : nullptr;
    rc = rerank_tail(ta, stream, tm, stats, 3);
    if (rc) return rc;
    if (ss && timing && stats) {   // the side stream's own duration (it overlaps the main chain)
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ss->t0, ss->t1) == hipSuccess) stats->ms_dq = ms;
    }
    // status of the data-dependent capacities (read with the statistics, after the fact): the fp16 operands must
    // not have overflowed and the fallback rows must have fitted their buffer -- otherwise the result above is not
    // valid and the caller repeats the call with the dense algorithm
    float gs[2] = {0.f, 0.f};
    unsigned fbc = 0;
    HIP_TRY(hipMemcpy(gs, gstat, 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&fbc, fb_count, 4, hipMemcpyDeviceToHost));
    if (gs[1] != 0.0f || fbc > (unsigned)L.fb_max) {
        mpreid_set_error("sparse re-ranking: %s; retry with MPREID_RERANK_DENSE",
                         gs[1] != 0.0f ? "row norms too large for fp16 operands" : "too many rows needed the dense fallback");
        return MPREID_ERR_RETRY_DENSE;
    }
    return MPREID_OK;
}

extern "C" size_t mpreid_rerank_workspace_bytes_ex(int64_t nq, int64_t ng, int d, int k1, int k2, int has_local, int algo) {
    if (nq < 0 || ng < 0 || d <= 0 || k1 < 0 || k2 < 1) return 0;
    const bool sparse = algo == MPREID_RERANK_SPARSE || algo == MPREID_RERANK_SPARSE_SPLIT3 ||
                        (algo == MPREID_RERANK_AUTO && sparse_eligible(nq, ng, k1, k2, has_local ? (const float *)1 : nullptr));
    return sparse ? make_layout2(nq, ng, d, k1, k2, algo == MPREID_RERANK_SPARSE_SPLIT3).total
                  : make_layout(nq, ng, d, k1, k2, has_local).total;
}
// The plain pair (mpreid_rerank_workspace_bytes + mpreid_rerank_f32: the binding INTEGRATION.md shows) is the DENSE
// algorithm: any N, local_distmat supported, no data-dependent MPREID_ERR_RETRY_DENSE -- it never fails on data.  The
// faster sparse algorithm (and the retry protocol that goes with it) is opt-in through the _ex entry points.
extern "C" size_t mpreid_rerank_workspace_bytes(int64_t nq, int64_t ng, int d, int k1, int k2, int has_local) {
    return mpreid_rerank_workspace_bytes_ex(nq, ng, d, k1, k2, has_local, MPREID_RERANK_DENSE);
}

extern "C" int mpreid_rerank_f32_ex(const float *q, const float *g, int64_t nq, int64_t ng, int d, int k1, int k2,
                                    double lambda_value, const float *local, int only_local, float *out, int64_t ldo,
                                    void *ws, size_t ws_bytes, mpreid_stream_t stream, mpreid_rerank_stats *stats,
                                    int timing, int algo) {
    ARG_CHECK(algo == MPREID_RERANK_AUTO || algo == MPREID_RERANK_DENSE || algo == MPREID_RERANK_SPARSE ||
              algo == MPREID_RERANK_SPARSE_SPLIT3);
    const bool eligible = sparse_eligible(nq, ng, k1, k2, local);
    const bool want_sparse = algo == MPREID_RERANK_SPARSE || algo == MPREID_RERANK_SPARSE_SPLIT3;
    if (want_sparse && !eligible) {
        mpreid_set_error("the sparse re-ranking algorithm needs N >= 2048, max(k1+1, k2) <= 64 and no local_distmat");
        return MPREID_ERR_UNSUPPORTED;
    }
    if (want_sparse || (algo == MPREID_RERANK_AUTO && eligible))
        return rerank_sparse(q, g, nq, ng, d, k1, k2, lambda_value, out, ldo, ws, ws_bytes, stream, stats, timing,
                             algo == MPREID_RERANK_SPARSE_SPLIT3);
    return rerank_dense(q, g, nq, ng, d, k1, k2, lambda_value, local, only_local, out, ldo, ws, ws_bytes, stream, stats,
                        timing);
}
extern "C" int mpreid_rerank_f32(const float *q, const float *g, int64_t nq, int64_t ng, int d, int k1, int k2,
                                 double lambda_value, const float *local, int only_local, float *out, int64_t ldo,
                                 void *ws, size_t ws_bytes, mpreid_stream_t stream, mpreid_rerank_stats *stats,
                                 int timing) {
    return mpreid_rerank_f32_ex(q, g, nq, ng, d, k1, k2, lambda_value, local, only_local, out, ldo, ws, ws_bytes, stream,
                                stats, timing, MPREID_RERANK_DENSE);
}

extern "C" int mpreid_rerank_debug_copy(const void *ws, int64_t nq, int64_t ng, int d, int k1, int k2, int has_local,
                                        int32_t *rank_out, int32_t *v_cnt, int32_t *vqe_cnt,
                                        mpreid_stream_t stream_) {
    return mpreid_rerank_debug_copy_ex(ws, nq, ng, d, k1, k2, has_local, rank_out, v_cnt, vqe_cnt, stream_,
                                       MPREID_RERANK_DENSE);
}

extern "C" int mpreid_rerank_debug_copy_ex(const void *ws, int64_t nq, int64_t ng, int d, int k1, int k2, int has_local,
                                           int32_t *rank_out, int32_t *v_cnt, int32_t *vqe_cnt,
                                           mpreid_stream_t stream_, int algo) {
    ARG_CHECK(ws && (algo == MPREID_RERANK_DENSE || algo == MPREID_RERANK_SPARSE || algo == MPREID_RERANK_SPARSE_SPLIT3));
    // the fields the taps read, from whichever layout the call used
    struct { int64_t N; int K, KR; size_t rank, vcnt, qcnt; } L;
    if (algo != MPREID_RERANK_DENSE) {
        const Rerank2Layout s2 = make_layout2(nq, ng, d, k1, k2);
        L.N = s2.N; L.K = s2.K; L.KR = s2.KR; L.rank = s2.rank; L.vcnt = s2.vcnt; L.qcnt = s2.qcnt;
    } else {
        const RerankLayout s1 = make_layout(nq, ng, d, k1, k2, has_local);
        L.N = s1.N; L.K = s1.K; L.KR = s1.KR; L.rank = s1.rank; L.vcnt = s1.vcnt; L.qcnt = s1.qcnt;
    }
    hipStream_t stream = (hipStream_t)stream_;
    const char *base = (const char *)ws;
    if (rank_out) { // [N][k1+1]; columns past min(k1+1, N) are -1
        for (int64_t t = 0; t < L.N * (int64_t)(k1 + 1); ++t) rank_out[t] = -1;
        HIP_TRY(hipMemcpy2DAsync(rank_out, (size_t)(k1 + 1) * 4, base + L.rank, (size_t)L.KR * 4, (size_t)L.K * 4,
                                 (size_t)L.N, hipMemcpyDeviceToHost, stream));
    }
    if (v_cnt) HIP_TRY(hipMemcpyAsync(v_cnt, base + L.vcnt, (size_t)L.N * 4, hipMemcpyDeviceToHost, stream));
    if (vqe_cnt)
        HIP_TRY(hipMemcpyAsync(vqe_cnt, base + (k2 != 1 ? L.qcnt : L.vcnt), (size_t)L.N * 4, hipMemcpyDeviceToHost,
                               stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return MPREID_OK;
}

// ---------------------------------------------------------------------------------------------
// Row-sharded re-ranking: the same kernels driven phase by phase over a row range
// [r_lo, r_lo + rows) of the N x N problem.  Between the phases the caller all-gathers the rank table, the
// sparse V rows and the sparse V_qe rows (mpreid/distributed.py: RCCL through torch.distributed).  No
// floating-point reduction crosses ranks and every row is computed by exactly the same instruction
// sequence as on one GPU, so the result does not depend on the number of ranks.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rowmax_kernel(const float *__restrict__ M, int64_t ld, int64_t N,
                                                     float *__restrict__ rowmax) {
    __shared__ float s_red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *row = M + (int64_t)blockIdx.x * ld;
    float mx = -3.402823466e+38f;
    for (int j = tid; j < (int)N; j += 256) mx = fmaxf(mx, row[j]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    if (lane == 0) s_red[wave] = mx;
    __syncthreads();
    if (tid == 0) rowmax[blockIdx.x] = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
}

// copy ELL rows from row stride `ss` to row stride `ds` (ds >= max count); entries past the count are zeroed
__global__ __launch_bounds__(256) void pack_rows_kernel(const int *__restrict__ cnt, const int *__restrict__ idx,
                                                        const uint16_t *__restrict__ val, int64_t rows, int ss, int ds,
                                                        int *__restrict__ idx_out, uint16_t *__restrict__ val_out) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int c = cnt[r];
    for (int a = lane; a < ds; a += 64) {
        idx_out[r * ds + a] = a < c ? idx[r * ss + a] : 0;
        val_out[r * ds + a] = a < c ? val[r * ss + a] : (uint16_t)0;
    }
}

// phase 1: rows [r_lo, r_lo+rows) of D = |f_i|^2 + |f_j|^2 - 2 f_i.f_j  (norms_all from mpreid_sqnorm_f32),
// their row maxima and (if rank_local != NULL) their first KR neighbours.  D_local [rows][ld], ld >= N.
extern "C" int mpreid_rr_dist_rows(const float *feat_all, const float *norms_all, int64_t n, int d, int64_t r_lo,
                                   int64_t rows, float *d_local, int64_t ld, float *rowmax_local, int32_t *rank_local,
                                   int kr, mpreid_stream_t stream_) {
    ARG_CHECK(feat_all && norms_all && d_local && rowmax_local && n > 0 && d > 0 && rows > 0 && r_lo >= 0 &&
              r_lo + rows <= n && ld >= n);
    hipStream_t stream = (hipStream_t)stream_;
    int rc = mpreid_distance_launch(feat_all + r_lo * (int64_t)d, feat_all, rows, n, d, norms_all + r_lo, norms_all,
                                    d_local, ld, 0, stream);
    if (rc) return rc;
    if (rank_local) {
        ARG_CHECK(kr >= 1 && kr <= 256 && kr <= n);
        rc = launch_rowmax_topk(d_local, ld, n, kr, rows, rowmax_local, rank_local, stream);
        if (rc) return rc;
    } else {
        hipLaunchKernelGGL(rowmax_kernel, dim3((unsigned)rows), dim3(256), 0, stream, d_local, ld, n, rowmax_local);
    }
    LAUNCH_CHECK();
    return MPREID_OK;
}

// phase 2: V rows of the local row range from the GLOBAL rank table; ELL with row stride vcap =
// min(N, (k1+1)*(1+half_k1)) (mpreid_rr_vcap)
extern "C" int mpreid_rr_vcap(int64_t n, int k1) {
    const int64_t K = std::min<int64_t>(k1 + 1, n), h = std::min<int64_t>(mpreid_half_k1(k1), n);
    const int64_t cap = K * (1 + h);
    return (int)(cap < n ? cap : n);
}

extern "C" size_t mpreid_rr_krecip_scratch_bytes(int64_t n) { return (size_t)n * RB_WORDS * 4 * 2; }

extern "C" int mpreid_rr_krecip(const float *d_local, int64_t ld, int64_t n, const float *rowmax_local,
                                const int32_t *rank_all, int k1, int kr, int64_t r_lo, int64_t rows, int32_t *vcnt,
                                int32_t *vidx, uint16_t *vval, void *scratch, mpreid_stream_t stream_) {
    const int K = (int)std::min<int64_t>(k1 + 1, n), h = (int)std::min<int64_t>(mpreid_half_k1(k1), n);
    const int vcap = mpreid_rr_vcap(n, k1);
    ARG_CHECK(d_local && rowmax_local && rank_all && vcnt && vidx && vval && scratch && rows > 0 && kr >= K);
    const int nw = (int)((n + 31) >> 5);
    const size_t lds = krecip_lds_bytes(false, nw, K, vcap, 0);
    int rc = set_krecip_lds(krecip_kernel<false>, lds, K, n);
    if (rc) return rc;
    // reciprocity bits of ALL rows (candidates of a local row live anywhere): recomputed by every rank from the
    // all-gathered table -- N x K membership tests, cheaper than another exchange
    unsigned *rk = (unsigned *)scratch;
    hipLaunchKernelGGL(recip_bits_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream_, rank_all, n, K, kr,
                       h, rk, rk + (size_t)n * RB_WORDS);
    hipLaunchKernelGGL(krecip_kernel<false>, dim3((unsigned)rows), dim3(256), lds, (hipStream_t)stream_, d_local, ld, n,
                       rowmax_local, rank_all, K, kr, h, vcap, vcnt, vidx, vval, (int)r_lo,
                       (int *)nullptr, (const float *)nullptr, (const float *)nullptr, 0, (const float *)nullptr, rk,
                       rk + (size_t)n * RB_WORDS);
    LAUNCH_CHECK();
    return MPREID_OK;
}

// ---------------------------------------------------------------------------------------------
// Row-sharded SPARSE phases (the multi-GPU form of the candidate pipeline): a rank's rows [r_lo, r_lo + rows) of the
// N x N problem without its [rows][N] distance block.
//   mpreid_rr_neighbours_sparse   phase 1: fp16 operands of all rows (every rank needs all columns), sample pass and
//                                 thresholds for the local rows, candidate GEMM local rows x all columns (no symmetry
//                                 to exploit inside a row block), exact refinement + fallback rows
//                                 -> rank_local [rows][kr], rowmax_local [rows], rankd_local [rows][kr]
//   mpreid_rr_krecip_sparse       phase 2 with on-the-fly exact distances (no D rows)
// Phases 3 and 4 are the existing ones (the Jaccard phase computes the exact rows of the LOCAL QUERIES only).  Same
// bits as the dense phases and the single-GPU call.  MPREID_ERR_RETRY_DENSE: use mpreid_rr_dist_rows for this rank.
// ---------------------------------------------------------------------------------------------
struct RrSparseLayout {
    int64_t Np, Nsp, Mp, ld, fb_max;
    int dp;
    size_t sqn, feat16, samp16, sampn, sampD, tlo, thi, eps, cnt_lo, cnt_hi, list_lo, list_hi, gstat, fb_count, fb_rows, fb_feat,
        fb_sqn, fb_D, fb_rowmax, fb_rank, total;
};
static RrSparseLayout rr_sparse_layout(int64_t n, int d, int64_t rows, int kr) {
    RrSparseLayout L{};
    L.Np = (int64_t)align_up((size_t)n, 256);
    L.Nsp = (int64_t)align_up((size_t)((n + RR2_SAMPLE_STRIDE - 1) / RR2_SAMPLE_STRIDE), 256);
    L.Mp = (int64_t)align_up((size_t)rows, 256);
    L.ld = (int64_t)align_up((size_t)n, 64);
    L.dp = (int)align_up((size_t)d, 64);
    L.fb_max = std::max<int64_t>(256, rows / 16);
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += align_up(bytes, 256);
        return o;
    };
    const size_t Np = (size_t)L.Np, Mp = (size_t)L.Mp, F = (size_t)L.fb_max;
    L.sqn = take((Np + 256) * 4);                      // + slack: the local row block is read in whole 256-row tiles
    L.feat16 = take((Np + 256) * (size_t)L.dp * 2);
    L.samp16 = take((size_t)L.Nsp * L.dp * 2);
    L.sampn = take((size_t)L.Nsp * 4);
    L.sampD = take(Mp * (size_t)L.Nsp * 4);
    L.tlo = take(Mp * 4);
    L.thi = take(Mp * 4);
    L.eps = take(Mp * 4);
    L.cnt_lo = take(Mp * 4);
    L.cnt_hi = take(Mp * 4);
    L.list_lo = take(Mp * (size_t)RR2_CAP_LO * 8);
    L.list_hi = take(Mp * (size_t)RR2_CAP_HI * 8);
    L.gstat = take(64);
    L.fb_count = take(64);
    L.fb_rows = take(F * 4);
    L.fb_feat = take(F * (size_t)d * 4);
    L.fb_sqn = take(F * 4);
    L.fb_D = take(F * (size_t)L.ld * 4);
    L.fb_rowmax = take(F * 4);
    L.fb_rank = take(F * (size_t)kr * 4);
    L.total = off;
    return L;
}

extern "C" size_t mpreid_rr_sparse_workspace_bytes(int64_t n, int d, int64_t rows, int kr) {
    if (n <= 0 || d <= 0 || rows <= 0 || kr <= 0) return 0;
    return rr_sparse_layout(n, d, rows, kr).total;
}

extern "C" int mpreid_rr_neighbours_sparse(const float *feat_all, const float *norms_all, int64_t n, int d, int64_t r_lo,
                                           int64_t rows, int kr, int32_t *rank_local, float *rowmax_local, float *rankd_local,
                                           void *ws, size_t ws_bytes, mpreid_stream_t stream_) {
    ARG_CHECK(feat_all && norms_all && rank_local && rowmax_local && rankd_local && n >= 2048 && d > 0 && rows > 0 && r_lo >= 0 &&
              r_lo + rows <= n && kr >= 1 && kr <= 64 && kr <= n);
    const RrSparseLayout L = rr_sparse_layout(n, d, rows, kr);
    if (!ws || ws_bytes < L.total) {
        mpreid_set_error("sparse phase-1 workspace too small: %zu < %zu", ws_bytes, L.total);
        return MPREID_ERR_WORKSPACE;
    }
    hipStream_t stream = (hipStream_t)stream_;
    char *base = (char *)ws;
    float *sqn = (float *)(base + L.sqn);
    _Float16 *feat16 = (_Float16 *)(base + L.feat16), *samp16 = (_Float16 *)(base + L.samp16);
    float *sampn = (float *)(base + L.sampn), *sampD = (float *)(base + L.sampD);
    float *tlo = (float *)(base + L.tlo), *thi = (float *)(base + L.thi), *eps = (float *)(base + L.eps);
    unsigned *cnt_lo = (unsigned *)(base + L.cnt_lo), *cnt_hi = (unsigned *)(base + L.cnt_hi);
    uint2 *list_lo = (uint2 *)(base + L.list_lo), *list_hi = (uint2 *)(base + L.list_hi);
    float *gstat = (float *)(base + L.gstat);
    unsigned *fb_count = (unsigned *)(base + L.fb_count);
    int *fb_rows = (int *)(base + L.fb_rows);
    float *fb_feat = (float *)(base + L.fb_feat), *fb_sqn = (float *)(base + L.fb_sqn), *fb_D = (float *)(base + L.fb_D),
          *fb_rowmax = (float *)(base + L.fb_rowmax);
    int *fb_rank = (int *)(base + L.fb_rank);
    const int ns = (int)((n + RR2_SAMPLE_STRIDE - 1) / RR2_SAMPLE_STRIDE);

    HIP_TRY(hipMemsetAsync(fb_count, 0, 64, stream));
    HIP_TRY(hipMemsetAsync(rank_local, 0, (size_t)rows * kr * 4, stream));
    HIP_TRY(hipMemsetAsync(rankd_local, 0, (size_t)rows * kr * 4, stream));
    HIP_TRY(hipMemsetAsync(rowmax_local, 0, (size_t)rows * 4, stream));
    HIP_TRY(hipMemsetAsync(sqn, 0, (size_t)(L.Np + 256) * 4, stream));
    HIP_TRY(hipMemcpyAsync(sqn, norms_all, (size_t)n * 4, hipMemcpyDeviceToDevice, stream));
    hipLaunchKernelGGL(rr2_norm_stats_kernel, dim3(1), dim3(1024), 0, stream, sqn, n, gstat);
    hipLaunchKernelGGL(rr2_cast_rows_kernel, dim3((unsigned)(L.Np + 256)), dim3(256), 0, stream, feat_all, sqn, n, d, 1, feat16,
                       L.dp, (float *)nullptr);
    hipLaunchKernelGGL(rr2_cast_rows_kernel, dim3((unsigned)L.Nsp), dim3(256), 0, stream, feat_all, sqn, n, d, RR2_SAMPLE_STRIDE,
                       samp16, L.dp, sampn);
    LAUNCH_CHECK();
    int rc;
    {   // sample pass for the local rows
        GemmArgs a{};
        a.A = feat16 + (size_t)r_lo * L.dp; a.W = samp16; a.M = (int)L.Mp; a.N = (int)L.Nsp; a.K = L.dp;
        a.out = sampD; a.ldo = L.Nsp; a.aux = sqn + r_lo; a.aux2 = sampn; a.m_valid = (int)rows; a.n_valid = ns;
        if ((rc = launch_gemm_f16(a, GE_EUCLID, stream))) return rc;
    }
    hipLaunchKernelGGL(rr2_threshold_kernel, dim3((unsigned)L.Mp), dim3(256), 0, stream, sampD, L.Nsp, ns, rows, RR2_RSEL,
                       sqn + r_lo, gstat, d, tlo, thi, eps, cnt_lo, cnt_hi);
    LAUNCH_CHECK();
    {   // candidates: local rows x all columns
        GemmArgs a{};
        a.A = feat16 + (size_t)r_lo * L.dp; a.W = feat16; a.M = (int)L.Mp; a.N = (int)L.Np; a.K = L.dp;
        a.aux = sqn + r_lo; a.aux2 = sqn; a.m_valid = (int)rows; a.n_valid = (int)n;
        a.tlo = tlo; a.thi = thi; a.cnt_lo = cnt_lo; a.cnt_hi = cnt_hi; a.list_lo = list_lo; a.list_hi = list_hi;
        a.cap_lo = RR2_CAP_LO; a.cap_hi = RR2_CAP_HI; a.sym = 0;
        if ((rc = launch_gemm_f16(a, GE_CAND, stream))) return rc;
    }
    {
        const size_t lds = 512 * 8 + RR2_MAXE * 8 + RR2_MAXH * 8 + (size_t)((d + 3) & ~3) * 4 + 2 * 64 * WXD_STRIDE * 4;
        if ((rc = set_dyn_lds(rr2_refine_kernel<RR2_CAP_LO, RR2_CAP_HI>, lds))) return rc;
        hipLaunchKernelGGL((rr2_refine_kernel<RR2_CAP_LO, RR2_CAP_HI>), dim3((unsigned)rows), dim3(256), lds, stream, feat_all, sqn,
                           n, d, kr, cnt_lo, list_lo, cnt_hi, list_hi, tlo, eps, rowmax_local, rank_local, rankd_local, fb_count,
                           fb_rows, (unsigned)L.fb_max, r_lo);
        hipLaunchKernelGGL(rr2_fb_gather_kernel, dim3((unsigned)L.fb_max), dim3(256), 0, stream, feat_all, sqn, d, fb_rows, fb_count,
                           (unsigned)L.fb_max, fb_feat, fb_sqn, r_lo);
        LAUNCH_CHECK();
        if ((rc = mpreid_distance_launch(fb_feat, feat_all, L.fb_max, n, d, fb_sqn, sqn, fb_D, L.ld, 0, stream, fb_count))) return rc;
        if ((rc = launch_rowmax_topk(fb_D, L.ld, n, kr, L.fb_max, fb_rowmax, fb_rank, stream, fb_count))) return rc;
        hipLaunchKernelGGL(rr2_fb_scatter_kernel, dim3((unsigned)L.fb_max), dim3(256), 0, stream, fb_rows, fb_count,
                           (unsigned)L.fb_max, fb_rowmax, fb_rank, kr, fb_D, L.ld, rowmax_local, rank_local, rankd_local);
        LAUNCH_CHECK();
    }
    float gs[2] = {0.f, 0.f};
    unsigned fbc = 0;
    HIP_TRY(hipMemcpyAsync(gs, gstat, 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(&fbc, fb_count, 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (gs[1] != 0.0f || fbc > (unsigned)L.fb_max) {
        mpreid_set_error("sparse phase 1: %s; use mpreid_rr_dist_rows for this row block",
                         gs[1] != 0.0f ? "row norms too large for fp16 operands" : "too many rows needed the dense fallback");
        return MPREID_ERR_RETRY_DENSE;
    }
    return MPREID_OK;
}

extern "C" int mpreid_rr_krecip_sparse(const float *feat_all, const float *norms_all, int64_t n, int d,
                                       const float *rowmax_local, const int32_t *rank_all, const float *rankd_local, int k1,
                                       int kr, int64_t r_lo, int64_t rows, int32_t *vcnt, int32_t *vidx, uint16_t *vval,
                                       void *scratch, mpreid_stream_t stream_) {
    const int K = (int)std::min<int64_t>(k1 + 1, n), h = (int)std::min<int64_t>(mpreid_half_k1(k1), n);
    const int vcap = mpreid_rr_vcap(n, k1);
    ARG_CHECK(feat_all && norms_all && rowmax_local && rank_all && rankd_local && vcnt && vidx && vval && scratch && rows > 0 &&
              kr >= K);
    const int nw = (int)((n + 31) >> 5);
    const size_t lds = krecip_lds_bytes(true, nw, K, vcap, d);
    int rc = set_krecip_lds(krecip_kernel<true>, lds, K, n);
    if (rc) return rc;
    unsigned *rk = (unsigned *)scratch;
    hipLaunchKernelGGL(recip_bits_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream_, rank_all, n, K, kr,
                       h, rk, rk + (size_t)n * RB_WORDS);
    // rankd is indexed by the global row inside the kernel: shift the local table's base accordingly
    hipLaunchKernelGGL(krecip_kernel<true>, dim3((unsigned)rows), dim3(256), lds, (hipStream_t)stream_, (const float *)nullptr,
                       (int64_t)0, n, rowmax_local, rank_all, K, kr, h, vcap, vcnt, vidx, vval, (int)r_lo, (int *)nullptr, feat_all,
                       norms_all, d, rankd_local - (int64_t)r_lo * kr, rk, rk + (size_t)n * RB_WORDS);
    LAUNCH_CHECK();
    return MPREID_OK;
}

extern "C" int mpreid_rr_pack_rows(const int32_t *cnt, const int32_t *idx, const uint16_t *val, int64_t rows,
                                   int src_stride, int dst_stride, int32_t *idx_out, uint16_t *val_out,
                                   mpreid_stream_t stream_) {
    ARG_CHECK(cnt && idx && val && idx_out && val_out && rows > 0 && src_stride > 0 && dst_stride > 0);
    hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream_, cnt, idx,
                       val, rows, src_stride, dst_stride, idx_out, val_out);
    LAUNCH_CHECK();
    return MPREID_OK;
}

// ---- CSR transport of the sparse rows (the all-gathers of V and V_qe move nnz entries instead of rows x global max) ----
// rowptr[0 .. rows] = exclusive prefix sums of cnt (one workgroup walks the rows in 1024-row chunks with a running carry)
__global__ __launch_bounds__(1024) void rowptr_scan_kernel(const int *__restrict__ cnt, int64_t rows, long long *__restrict__ rowptr) {
    __shared__ long long wsum[16];
    __shared__ long long carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < rows; base += 1024) {
        const int64_t i = base + tid;
        const long long v = i < rows ? (long long)cnt[i] : 0;
        long long x = v;   // inclusive scan inside the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const long long y = __shfl_up(x, off, 64);
            if (lane >= off) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        long long pre = carry_s;
        for (int w = 0; w < wave; ++w) pre += wsum[w];
        if (i < rows) rowptr[i] = pre + x - v;
        __syncthreads();
        if (tid == 1023) carry_s = pre + x;
        __syncthreads();
    }
    if (tid == 0) rowptr[rows] = carry_s;
}

// TO_CSR: packed[rowptr[r] + e] = ell[r][e] for e < cnt[r]; else the inverse (ELL padding is left untouched)
template <bool TO_CSR>
__global__ __launch_bounds__(256) void csr_ell_kernel(const long long *__restrict__ rowptr, int64_t rows, int stride,
                                                      int *__restrict__ ell_idx, uint16_t *__restrict__ ell_val,
                                                      int *__restrict__ csr_idx, uint16_t *__restrict__ csr_val) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const long long p0 = rowptr[r];
    const int n = (int)(rowptr[r + 1] - p0);
    for (int e = threadIdx.x & 63; e < n; e += 64) {
        if (TO_CSR) {
            csr_idx[p0 + e] = ell_idx[r * stride + e];
            csr_val[p0 + e] = ell_val[r * stride + e];
        } else {
            ell_idx[r * stride + e] = csr_idx[p0 + e];
            ell_val[r * stride + e] = csr_val[p0 + e];
        }
    }
}

extern "C" int mpreid_rr_rowptr(const int32_t *cnt, int64_t rows, long long *rowptr, mpreid_stream_t stream_) {
    ARG_CHECK(cnt && rowptr && rows >= 0);
    hipLaunchKernelGGL(rowptr_scan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream_, cnt, rows, rowptr);
    LAUNCH_CHECK();
    return MPREID_OK;
}

extern "C" int mpreid_rr_ell_to_csr(const long long *rowptr, const int32_t *idx, const uint16_t *val, int64_t rows, int stride,
                                    int32_t *idx_out, uint16_t *val_out, mpreid_stream_t stream_) {
    ARG_CHECK(rowptr && idx && val && idx_out && val_out && rows >= 0 && stride > 0);
    if (rows == 0) return MPREID_OK;
    hipLaunchKernelGGL(csr_ell_kernel<true>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream_, rowptr, rows,
                       stride, const_cast<int *>(idx), const_cast<uint16_t *>(val), idx_out, val_out);
    LAUNCH_CHECK();
    return MPREID_OK;
}

extern "C" int mpreid_rr_csr_to_ell(const long long *rowptr, const int32_t *idx, const uint16_t *val, int64_t rows, int stride,
                                    int32_t *idx_out, uint16_t *val_out, mpreid_stream_t stream_) {
    ARG_CHECK(rowptr && idx && val && idx_out && val_out && rows >= 0 && stride > 0);
    if (rows == 0) return MPREID_OK;
    hipLaunchKernelGGL(csr_ell_kernel<false>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream_, rowptr, rows,
                       stride, idx_out, val_out, const_cast<int *>(idx), const_cast<uint16_t *>(val));
    LAUNCH_CHECK();
    return MPREID_OK;
}

// phase 3: query expansion of the local rows from the GLOBAL V (row stride vstride).  Two steps: union sizes
// (ucnt_local), then the fill with row stride qcap >= max union size.
extern "C" int mpreid_rr_qe_count(int64_t n, const int32_t *rank_all, int kr, int k2, int64_t r_lo, int64_t rows,
                                  const int32_t *vcnt_all, const int32_t *vidx_all, int vstride, int32_t *ucnt_local,
                                  mpreid_stream_t stream_) {
    ARG_CHECK(rank_all && vcnt_all && vidx_all && ucnt_local && rows > 0 && k2 >= 1 && kr >= k2);
    const size_t lds = (size_t)((n + 31) >> 5) * 4;
    int rc = set_dyn_lds(qe_count_kernel, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(qe_count_kernel, dim3((unsigned)rows), dim3(64), lds, (hipStream_t)stream_, n, rank_all, kr, k2,
                       vcnt_all, vidx_all, vstride, ucnt_local, (int)r_lo);
    LAUNCH_CHECK();
    return MPREID_OK;
}

extern "C" int mpreid_rr_qe_fill(int64_t n, const int32_t *rank_all, int kr, int k2, int64_t r_lo, int64_t rows,
                                 const int32_t *vcnt_all, const int32_t *vidx_all, const uint16_t *vval_all, int vstride,
                                 int qcap, int32_t *qcnt_local, int32_t *qidx_local, uint16_t *qval_local,
                                 mpreid_stream_t stream_) {
    ARG_CHECK(rank_all && vcnt_all && vidx_all && vval_all && qcnt_local && qidx_local && qval_local && rows > 0 &&
              qcap > 0);
    const size_t lds = (size_t)((n + 31) >> 5) * 8 + (size_t)qcap * 8;
    int rc = set_dyn_lds(qe_fill_kernel, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(qe_fill_kernel, dim3((unsigned)rows), dim3(64), lds, (hipStream_t)stream_, n, rank_all, kr, k2,
                       vcnt_all, vidx_all, vval_all, vstride, qcap, qcnt_local, qidx_local, qval_local, (int)r_lo);
    LAUNCH_CHECK();
    return MPREID_OK;
}

// phase 4: inverted index of the GLOBAL V_qe (row stride qstride) + Jaccard / blend for the query rows
// [q_lo, q_lo + qrows): d_q [qrows][ld] and rowmax_q from mpreid_rr_dist_rows on that range.
// Scratch: ccnt [N+1] u32, cptr [N+1] i64, crow [nnz] i32, cval [nnz] u16 (nnz = sum of qcnt_all).
extern "C" size_t mpreid_rr_jaccard_hist_bytes(int64_t n) { return csc_hist_bytes(n); }

extern "C" int mpreid_rr_jaccard(int64_t n, int64_t nq, int64_t q_lo, int64_t qrows, const float *d_q, int64_t ld,
                                 const float *rowmax_q, const int32_t *qcnt_all, const int32_t *qidx_all,
                                 const uint16_t *qval_all, int qstride, double lambda_value, uint32_t *ccnt,
                                 long long *cptr, int32_t *crow, uint16_t *cval, uint32_t *chist, float *out, int64_t ldo,
                                 mpreid_stream_t stream_) {
    ARG_CHECK(d_q && rowmax_q && qcnt_all && qidx_all && qval_all && ccnt && cptr && crow && cval && out && qrows > 0 &&
              q_lo >= 0 && q_lo + qrows <= nq && ldo >= n - nq);
    hipStream_t stream = (hipStream_t)stream_;
    bool blocked = false;
    int rc = launch_csc(n, nq, qcnt_all, qidx_all, qval_all, qstride, ccnt, chist, cptr, crow, cval, stream, &blocked);
    if (rc) return rc;
    return launch_jaccard(n, nq, (int)q_lo, qrows, d_q, ld, rowmax_q, qcnt_all, qidx_all, qval_all, qstride, cptr, crow, cval,
                          chist, blocked, lambda_value, out, ldo, nullptr, stream);
}

// ---- column-sharded build of the inverted index (phase 4 of the row-sharded re-ranking on P ranks) -------------------------
// mpreid_rr_jaccard above builds the WHOLE index on every rank: the one part of the sharded re-ranking that does not shrink
// as 1/P.  Here rank r builds the columns [c_lo, c_hi) of its column shard only -- the same counting sort (rows cut into the
// same CSC_B blocks, the same packed entries and chunk-boundary table), restricted to a column window -- and the ranks
// exchange (1) the column counts [N] u32, (2) their contiguous piece of the packed index, (3) their rows of the boundary
// table.  Within a column the entries are grouped by row block exactly as in the single build; the order inside a block is
// LDS-atomic arrival order in both, and the Jaccard sum does not depend on it (every row occurs once per column and owns its
// accumulator): outputs are bit-identical (tests/test_gpu_rerank.py::test_sharded_sparse_phases_equal_single_call).
// (n, nq) of the three csc entry points: at least one INDEXED row (nq < n: jaccard_plan divides by the rows per block) and
// packed positions that fit 32 bits
static bool csc_shape_ok(int64_t n, int64_t nq) { return n > 0 && nq >= 0 && nq < n; }
extern "C" int mpreid_rr_csc_chunks(int64_t n, int64_t nq) {
    ARG_CHECK(csc_shape_ok(n, nq));
    return jaccard_plan(n - nq).nchunks;
}

// step 1: ccnt[c] = number of indexed entries (rows [nq, n)) of every column c in [c_lo, c_hi); chist (mpreid_rr_jaccard_hist_bytes)
// keeps the per-block offsets of those columns for step 2
extern "C" int mpreid_rr_csc_count(int64_t n, int64_t nq, const int32_t *qcnt_all, const int32_t *qidx_all, int qstride,
                                   int64_t c_lo, int64_t c_hi, uint32_t *chist, uint32_t *ccnt, mpreid_stream_t stream_) {
    ARG_CHECK(qcnt_all && qidx_all && chist && ccnt && qstride > 0 && 0 <= c_lo && c_lo <= c_hi && c_hi <= n && csc_shape_ok(n, nq));
    if ((uint64_t)n * (uint64_t)qstride >= (1ull << 32)) {
        mpreid_set_error("mpreid_rr_csc_count: n * qstride must be below 2^32 (packed positions)");
        return MPREID_ERR_ARG;
    }
    if (c_hi == c_lo) return MPREID_OK;
    hipStream_t stream = (hipStream_t)stream_;
    const JaccardPlan jp = jaccard_plan(n - nq);
    const int64_t cols = c_hi - c_lo;
    const int nranges = (int)((cols + CSC_CR - 1) / CSC_CR);
    const size_t lds = (size_t)std::min<int64_t>(cols, CSC_CR) * 4;
    int rc = set_dyn_lds(csc2_hist_kernel, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(csc2_hist_kernel, dim3(CSC_B, nranges), dim3(1024), lds, stream, n, qcnt_all, qidx_all, qstride, jp.rpb,
                       chist, nq, c_lo, c_hi);
    hipLaunchKernelGGL(csc2_colscan_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, stream, n, chist, ccnt, c_lo, c_hi);
    LAUNCH_CHECK();
    return MPREID_OK;
}

// step 2 (after the all-gather of the counts): cptr[0 .. n] = exclusive scan of ccnt_all (ccnt_all is zeroed: scratch), then the
// packed entries of the columns [c_lo, c_hi) at their GLOBAL positions cpk[cptr[c_lo] .. cptr[c_hi]) and the rows [c_lo, c_hi)
// of the boundary table hb [n][mpreid_rr_csc_chunks + 1].  chist: as left by step 1.
extern "C" int mpreid_rr_csc_fill(int64_t n, int64_t nq, const int32_t *qcnt_all, const int32_t *qidx_all,
                                  const uint16_t *qval_all, int qstride, int64_t c_lo, int64_t c_hi, uint32_t *ccnt_all,
                                  uint32_t *chist, long long *cptr, uint32_t *cpk, uint32_t *hb, mpreid_stream_t stream_) {
    ARG_CHECK(qcnt_all && qidx_all && qval_all && ccnt_all && chist && cptr && cpk && hb && qstride > 0 && 0 <= c_lo &&
              c_lo <= c_hi && c_hi <= n && csc_shape_ok(n, nq));
    if ((uint64_t)n * (uint64_t)qstride >= (1ull << 32)) {   // (cptr is consumed as u32 positions by the fill and the Jaccard stage)
        mpreid_set_error("mpreid_rr_csc_fill: n * qstride must be below 2^32 (packed positions)");
        return MPREID_ERR_ARG;
    }
    hipStream_t stream = (hipStream_t)stream_;
    const JaccardPlan jp = jaccard_plan(n - nq);
    {
        const int nt = (int)((n + 1023) / 1024);
        unsigned long long *tsum = (unsigned long long *)((char *)chist + align_up((size_t)n * (CSC_B + 257) * 4, 8));
        hipLaunchKernelGGL(scan_tile_sums_kernel, dim3((unsigned)nt), dim3(256), 0, stream, n, ccnt_all, tsum);
        hipLaunchKernelGGL(scan_tile_bases_kernel, dim3(1), dim3(1024), 0, stream, nt, tsum, cptr + n);
        hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)nt), dim3(256), 0, stream, n, ccnt_all, tsum, cptr);
    }
    if (c_hi > c_lo) {
        const int64_t cols = c_hi - c_lo;
        const int nranges = (int)((cols + CSC_CR - 1) / CSC_CR);
        const size_t lds = (size_t)std::min<int64_t>(cols, CSC_CR) * 4;
        int rc = set_dyn_lds(csc2_fill_kernel, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(csc2_fill_kernel, dim3(CSC_B, nranges), dim3(1024), lds, stream, n, qcnt_all, qidx_all, qval_all, qstride,
                           jp.rpb, chist, cptr, cpk, nq, jp.bpc, c_lo, c_hi);
        hipLaunchKernelGGL(csc2_bounds_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, stream, n, jp.nchunks, jp.bpc,
                           chist, cptr, hb, c_lo, c_hi);
    }
    LAUNCH_CHECK();
    return MPREID_OK;
}

// step 3 (after the all-gathers of the index pieces and of the boundary rows): Jaccard + blend of this rank's queries over the
// assembled index -- the second half of mpreid_rr_jaccard
extern "C" int mpreid_rr_jaccard_indexed(int64_t n, int64_t nq, int64_t q_lo, int64_t qrows, const float *d_q, int64_t ld,
                                         const float *rowmax_q, const int32_t *qcnt_all, const int32_t *qidx_all,
                                         const uint16_t *qval_all, int qstride, double lambda_value, const long long *cptr,
                                         const uint32_t *cpk, const uint32_t *hb, float *out, int64_t ldo,
                                         mpreid_stream_t stream_) {
    ARG_CHECK(d_q && rowmax_q && qcnt_all && qidx_all && qval_all && cptr && cpk && hb && out && qrows > 0 && q_lo >= 0 &&
              q_lo + qrows <= nq && ldo >= n - nq);
    return launch_jaccard(n, nq, (int)q_lo, qrows, d_q, ld, rowmax_q, qcnt_all, qidx_all, qval_all, qstride, cptr,
                          (const int *)cpk, nullptr, nullptr, true, lambda_value, out, ldo, nullptr, (hipStream_t)stream_, hb);
}
