// common.h — shared host/device helpers of libmpreid_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <mutex>

#include "../../include/mpreid.h"
#include "../../include/mpreid_numerics.h"

void mpreid_set_error(const char *fmt, ...);

#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            mpreid_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return (int)e_;                                                                         \
        }                                                                                           \
    } while (0)
#define LAUNCH_CHECK() HIP_TRY(hipGetLastError())
#define ARG_CHECK(cond)                                                                             \
    do {                                                                                            \
        if (!(cond)) {                                                                              \
            mpreid_set_error("bad argument: %s (%s:%d)", #cond, __FILE__, __LINE__);               \
            return MPREID_ERR_ARG;                                                                  \
        }                                                                                           \
    } while (0)

__host__ __device__ static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Timing-ablation switches (MPREID_GEMM_DBG, MPREID_ATT_DBG, MPREID_CAND_DBG, MPREID_JACCARD_DBG: kernels that skip loads,
// matrix instructions or stores -- WRONG RESULTS by design) are honoured only by a library built with -DMPREID_ABLATION
// (MPREID_ABLATION=1 python mp-reid_amd/mpreid/build.py); the shipped library ignores them, so a stray environment
// variable cannot change results.
static inline int mpreid_ablation_env(const char *name) {
#ifdef MPREID_ABLATION
    const char *e = getenv(name);
    return e ? atoi(e) : 0;
#else
    (void)name;
    return 0;
#endif
}

// Tuning switches between BIT-IDENTICAL forms of a stage (kernel shape selection, side-stream overlap): ONE documented
// environment string, MPREID_TUNE="key=value,key=value" (include/mpreid.h lists the keys), read once per process.  A key that
// is absent returns `dflt`; unknown keys are reported once on stderr -- a stray or misspelt variable cannot silently change
// which kernel is measured.  (Switches that change RESULTS exist only in -DMPREID_ABLATION builds, above.)
int mpreid_tune(const char *key, int dflt);

// Bijective XCD-aware remap of a 1-D block id: blocks b and b+8 share an XCD (round-robin
// dispatch), so give each XCD a contiguous chunk of the logical tile order (L2 locality only;
// correctness never depends on it).  cdna_hip_programming.md §5 "XCD swizzle must be bijective".
__device__ __forceinline__ unsigned xcd_remap(unsigned orig, unsigned nwg) {
    const unsigned q = nwg >> 3, r = nwg & 7u, xcd = orig & 7u;
    const unsigned base = (xcd < r) ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return base + (orig >> 3);
}

// grouped tile order: walk GM tile-rows at a time so concurrently resident blocks share B panels
__device__ __forceinline__ void tile_coords(unsigned id, int tiles_m, int tiles_n, int GM, int &tm, int &tn) {
    const unsigned per_group = (unsigned)GM * (unsigned)tiles_n;
    const unsigned group = id / per_group;
    const int first_m = (int)group * GM;
    const int gsize = min(tiles_m - first_m, GM);
    const unsigned in_group = id % per_group;
    tm = first_m + (int)(in_group % (unsigned)gsize);
    tn = (int)(in_group / (unsigned)gsize);
}

// fp16 PAIR split of the `split` precision modes, x = hi + lo with hi = fp16(x), lo = fp16(x - float(hi)), two values at a
// time in THREE instructions: v_cvt_pk_f16_f32 (both hi), then v_fma_mixlo_f16 / v_fma_mixhi_f16 computing fma(hi, -1, x)
// from the packed hi halves and rounding it once into the low / high half of the packed lo.  x - hi is exact in fp32, so the
// single rounding gives the bits of the plain form (tools/probes/mix_probe.hip: 2^25 pairs incl. 14 M subnormal lo, 0
// mismatches); hipcc compiles the plain form to 6.5 instructions per two values (separate conversions back to fp32, subtract,
// convert, pack) and folds fma(hi, -1, x) back into a subtraction, hence the inline assembly.
typedef _Float16 mp_h2_t __attribute__((ext_vector_type(2)));
typedef float mp_f2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair2(float x0, float x1, mp_h2_t &hi, mp_h2_t &lo) {
    hi = __builtin_convertvector(mp_f2_t{x0, x1}, mp_h2_t);
    const unsigned hb = __builtin_bit_cast(unsigned, hi);
    unsigned d;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hb), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(hb), "v"(x1));
    lo = __builtin_bit_cast(mp_h2_t, d);
}
// N values (N even) -> vectors of N halves (any ext_vector_type of _Float16 with N elements)
template <int N, typename HV>
__device__ __forceinline__ void split_pairs(const float (&x)[N], HV &hi, HV &lo) {
    static_assert(N % 2 == 0, "pairs");
#pragma unroll
    for (int j = 0; j < N; j += 2) {
        mp_h2_t h, l;
        split_pair2(x[j], x[j + 1], h, l);
        hi[j] = h[0]; hi[j + 1] = h[1];
        lo[j] = l[0]; lo[j + 1] = l[1];
    }
}

__device__ __forceinline__ float wave_bfly_add(float s) {
    // fixed butterfly 32,16,8,4,2,1 — the order the oracle mirrors (oracle/mpreid_oracle.c sqnorm_row)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s = s + __shfl_xor(s, off, 64);
    return s;
}


// Per-launch event timing shared by the kernels bench.py reports a roofline for (gemm_f16.hip owns the registry).
// prof_begin returns an opaque token (nullptr while profiling is disabled); prof_end records the closing event and files
// the launch under (cls, m, n, k) with `work` = executed FLOPs (matrix kernels) or algorithmic bytes (HBM-bound kernels).
// cls: a GemmEpi id for the fp16 GEMM kernels, MPREID_PROF_* for the others.
void *mpreid_prof_begin(hipStream_t stream);
void mpreid_prof_end(void *token, hipStream_t stream, int cls, int64_t m, int n, int k, double work);

// Run `f` (returns an mpreid/hip status) once per HIP device for the call site that owns this object: function
// attributes (dynamic LDS size) and occupancy queries are per device, and one process may drive several GPUs.
struct PerDeviceOnce {
    std::mutex mu;
    bool done[64] = {};
    template <typename F>
    int run(F f) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        std::lock_guard<std::mutex> lk(mu);
        if (dev >= 0 && dev < 64 && done[dev]) return MPREID_OK;
        const int rc = f();
        if (rc == MPREID_OK && dev >= 0 && dev < 64) done[dev] = true;
        return rc;
    }
};
