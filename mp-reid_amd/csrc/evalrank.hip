// evalrank.hip — the ranking part of eval_func (reference utils/metrics.py:28-88) on the GPU.
//
// The reference argsorts every row of distmat (nq x ng) and derives CMC and AP from the positions of the
// gallery items that share the query's pid.  Only those positions are needed, so no sort is done here:
// per query (one 256-thread workgroup)
//   1. collect the relevant items' keys (distance, gallery index) and sort them (bitonic, LDS);
//   2. one pass over the row: every gallery item is dropped into the bucket between two consecutive
//      relevant keys (binary search, LDS histogram);
//   3. prefix sums give, for the t-th relevant item, its 0-based position in the full ascending
//      (value, index) order — exactly the position a stable argsort gives it.
// HBM-bound: 4*nq*ng bytes read once (+ pids); Market-1501 scale: 214 MB.
// The host finishes CMC / AP from the positions (float64, numpy's pairwise order) — utils/metrics.py.
#include "common.h"

constexpr int EV_CAP_MAX = 8192; // max relevant gallery items per query handled on the GPU (64 KB of keys + 32 KB of counters)
constexpr int EV_HIST_MIN = 2112; // counter words of the smallest instance: room for privatised copies of a short bucket list

__device__ __forceinline__ unsigned long long ev_key(float f, unsigned idx) {
    f = f + 0.0f; // -0 -> +0
    unsigned u = __float_as_uint(f);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | idx;
}

// pos_out [nq][rcap] int32 ascending positions (padded with -1), cnt_out [nq] (= -1 when the query has more
// than min(rcap, cap) relevant items: the caller falls back to the host for that row)
// Dynamic LDS: rel [cap] u64 | hist [hw] u32, cap a power of two >= 64, hw = max(cap + 1, EV_HIST_MIN).
// Step 2's counters are PRIVATISED: with R relevant items there are R + 1 buckets and C = the largest power of two with
// C * (R + 1) <= hw (at most 64) copies of them, copy = lane & (C - 1).  Round 5's single copy took one LDS atomic per
// gallery item into ~22 addresses (Market-1501: ~21 relevant items per query): up to 64 lanes of an instruction on one
// address, serialised by the LDS atomic unit; with R < 32 every lane owns its copy and no two lanes ever collide.
__global__ __launch_bounds__(256) void eval_rank_kernel(const float *__restrict__ dist, int64_t ld, int nq, int ng,
                                                        const long long *__restrict__ q_pids,
                                                        const long long *__restrict__ g_pids, int rcap, int cap, int hw,
                                                        int *__restrict__ pos_out, int *__restrict__ cnt_out) {
    extern __shared__ unsigned long long ev_lds[];
    unsigned long long *rel = ev_lds;
    unsigned *hist = reinterpret_cast<unsigned *>(ev_lds + cap);
    __shared__ unsigned s_cnt;
    __shared__ int s_wave[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = blockIdx.x;
    const float *row = dist + (int64_t)q * ld;
    const long long pid = q_pids[q];
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    // 1. relevant items (wave-aggregated reservation: one LDS atomic per wave and 64 items)
    for (int j0 = 0; j0 < ng; j0 += 256) {
        const int j = j0 + tid;
        const bool hit = j < ng && g_pids[j] == pid;
        const unsigned long long m = __ballot(hit);
        if (m) {
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(&s_cnt, (unsigned)__popcll(m));
            base = __shfl(base, 0, 64);
            if (hit) {
                const unsigned p = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
                if (p < (unsigned)cap) rel[p] = ev_key(row[j], (unsigned)j);
            }
        }
    }
    __syncthreads();
    const int R = (int)s_cnt;
    if (R > cap || R > rcap) {
        if (tid == 0) cnt_out[q] = -1;
        return;
    }
    if (tid == 0) cnt_out[q] = R;
    if (R == 0) {
        for (int t = tid; t < rcap; t += 256) pos_out[(int64_t)q * rcap + t] = -1;
        return;
    }
    int npow = 1;
    while (npow < R) npow <<= 1;
    for (int t = R + tid; t < npow; t += 256) rel[t] = ~0ull; // padding sorts last
    const int nb = R + 1;
    int C = 1;
    while (C < 64 && 2 * C * nb <= hw) C <<= 1;
    for (int t = tid; t < C * nb; t += 256) hist[t] = 0;
    __syncthreads();
    for (int size = 2; size <= npow; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < npow; t += 256) {
                const int partner = t ^ stride;
                if (partner > t) {
                    const unsigned long long a = rel[t], b = rel[partner];
                    const bool up = ((t & size) == 0);
                    if ((a > b) == up) {
                        rel[t] = b;
                        rel[partner] = a;
                    }
                }
            }
            __syncthreads();
        }
    // 2. bucket every gallery item: b = number of relevant keys < key_j (a relevant item t lands in bucket t)
    unsigned *mine = hist + (lane & (C - 1)) * nb;
    for (int j = tid; j < ng; j += 256) {
        const unsigned long long k = ev_key(row[j], (unsigned)j);
        int lo = 0, hi = R; // first t with rel[t] >= k
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (rel[mid] < k) lo = mid + 1; else hi = mid;
        }
        atomicAdd(&mine[lo], 1u); // items with key in (rel[lo-1], rel[lo]]
    }
    __syncthreads();
    if (C > 1) { // fold the copies into copy 0 (thread t touches column t of every copy and nothing else)
        for (int t = tid; t < nb; t += 256) {
            unsigned s = 0;
            for (int c = 0; c < C; ++c) s += hist[c * nb + t];
            hist[t] = s;
        }
        __syncthreads();
    }
    // 3. position of relevant item t = number of items with a smaller key = sum_{b<=t} hist[b] - 1 (itself)
    unsigned run = 0;
    for (int t0 = 0; t0 < R; t0 += 256) {
        const int t = t0 + tid;
        const int v = (t < R) ? (int)hist[t] : 0;
        // block inclusive scan
        int x = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int y = __shfl_up(x, off, 64);
            if (lane >= off) x += y;
        }
        __syncthreads();
        if (lane == 63) s_wave[wave] = x;
        __syncthreads();
        int base = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wave) base += s_wave[w];
            tot += s_wave[w];
        }
        if (t < R) pos_out[(int64_t)q * rcap + t] = (int)(run + (unsigned)(base + x)) - 1;
        run += (unsigned)tot;
        __syncthreads();
    }
    for (int t = R + tid; t < rcap; t += 256) pos_out[(int64_t)q * rcap + t] = -1;
}

extern "C" int mpreid_eval_rank_positions(const float *dist_dev, int64_t ld, int nq, int ng, const int64_t *q_pids_dev,
                                          const int64_t *g_pids_dev, int rcap, int32_t *pos_out_dev,
                                          int32_t *cnt_out_dev, mpreid_stream_t stream) {
    ARG_CHECK(dist_dev && q_pids_dev && g_pids_dev && pos_out_dev && cnt_out_dev && nq > 0 && ng > 0 && ld >= ng &&
              rcap > 0);
    int cap = 64;
    while (cap < rcap && cap < EV_CAP_MAX) cap <<= 1;
    const int hw = cap + 1 > EV_HIST_MIN ? cap + 1 : EV_HIST_MIN;
    const size_t lds = (size_t)cap * 8 + (size_t)hw * 4;
    static PerDeviceOnce attr_once;
    {
        const int rc = attr_once.run([]() -> int {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(eval_rank_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize,
                                        EV_CAP_MAX * 8 + (EV_CAP_MAX + 1) * 4));
            return MPREID_OK;
        });
        if (rc != MPREID_OK) return rc;
    }
    void *ptok = mpreid_prof_begin((hipStream_t)stream);
    hipLaunchKernelGGL(eval_rank_kernel, dim3((unsigned)nq), dim3(256), lds, (hipStream_t)stream, dist_dev, ld, nq, ng,
                       (const long long *)q_pids_dev, (const long long *)g_pids_dev, rcap, cap, hw, pos_out_dev,
                       cnt_out_dev);
    mpreid_prof_end(ptok, (hipStream_t)stream, MPREID_PROF_EVALRANK, nq, ng, 0, 4.0 * (double)nq * (double)ng);
    LAUNCH_CHECK();
    return MPREID_OK;
}
