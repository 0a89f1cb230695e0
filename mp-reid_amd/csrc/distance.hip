// distance.hip — L2-normalise, squared norms and the feat x feat^T distance GEMM (exact fp32 mode).
//
// Reference behaviour: utils/metrics.py:7-13 (euclidean_distance), :15-25 (cosine_similarity),
// :112-114 (F.normalize), utils/reranking.py:36-41 (all-pairs distance of cat(q,g)).
//
// Exact mode = v_mfma_f32_32x32x2_f32: per output element a k-ascending fp32 fmaf chain from 0
// (cdna_hip_programming.md §3 "FP32-input MFMA"), so D[i][j] is bit-identical to the oracle's
// scalar chain and D is bit-symmetric when q == g (the re-ranking path relies on that: column max
// == row max).  Roofline: MFMA fp32, 157 TFLOP/s peak.
//
// Tile: 128x128 per 256-thread workgroup (2x2 waves of 64x64 = 2x2 MFMA 32x32 tiles), BK = 16,
// LDS [k][m] fp32 double-buffered (33 KB -> 4 workgroups/CU), register-staged prefetch of the next
// K tile (global loads issued before the MFMA block, LDS writes after it), one barrier per K tile.
#include "common.h"

using f32x16 = __attribute__((ext_vector_type(16))) float;

// ---------------------------------------------------------------------------------------------
// squared norms / normalisation: one wave per row, lane-strided fmaf chains + fixed butterfly
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float row_sqnorm(const float *__restrict__ x, int d, int lane) {
    float acc = 0.0f;
    for (int k = lane; k < d; k += 64) {
        const float v = x[k];
        acc = fmaf(v, v, acc);
    }
    return wave_bfly_add(acc);
}

// mode 0: |x|^2, mode 1: |x|
__global__ __launch_bounds__(256) void sqnorm_kernel(const float *__restrict__ x, int64_t n, int d,
                                                     float *__restrict__ out, int mode) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    float s = row_sqnorm(x + row * (int64_t)d, d, lane);
    if (mode == 1) s = sqrtf(s);
    if (lane == 0) out[row] = s;
}

__global__ __launch_bounds__(256) void l2_normalize_kernel(const float *__restrict__ x, int64_t n, int d, float eps,
                                                           float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float *xr = x + row * (int64_t)d;
    float *orow = out + row * (int64_t)d;
    const float nrm = sqrtf(row_sqnorm(xr, d, lane)); // sqrtf = correctly rounded expansion; __fsqrt_rn is the bare v_sqrt_f32
    const float den = nrm > eps ? nrm : eps;
    for (int k = lane; k < d; k += 64) orow[k] = __fdiv_rn(xr[k], den);
}

// ---------------------------------------------------------------------------------------------
// exact fp32 MFMA GEMM with distance epilogues
// ---------------------------------------------------------------------------------------------
enum { EPI_EUCLID = 0, EPI_COSINE = 1,
       // linear layers of the all-fp32 encoder mode (vit.hip): bn = bias[n] (or NULL), C = out
       EPI_LIN = 2,        // C = acc + bias
       EPI_LIN_GELU = 3,   // C = quickgelu(acc + bias)
       EPI_LIN_RES = 4,    // C += acc + bias
       EPI_LIN_RELU = 5,   // C = relu(acc + bias)                      (RN50 fp32 mode: conv + folded BN + ReLU)
       EPI_LIN_RES_RELU = 6 };  // C = relu(C + (acc + bias))           (Bottleneck: relu(bn3(conv3) + identity), identity in C)

constexpr int XBM = 128, XBN = 128, XBK = 16, XLD = 132;
constexpr int XKC = XBK / 4;            // float4 chunks per row of a k tile
constexpr int XNR = XBM * XKC / 256;    // float4 per thread, operand and k tile

__device__ __forceinline__ float4 load_k4(const float *__restrict__ base, int64_t row, int64_t nrows, int k, int K,
                                          bool vec_ok) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < nrows) {
        const float *p = base + row * (int64_t)K + k;
        if (vec_ok && k + 3 < K) {
            v = *reinterpret_cast<const float4 *>(p);
        } else {
            if (k + 0 < K) v.x = p[0];
            if (k + 1 < K) v.y = p[1];
            if (k + 2 < K) v.z = p[2];
            if (k + 3 < K) v.w = p[3];
        }
    }
    return v;
}

// SYM: A == B (all-pairs distance of one feature set).  D is bit-symmetric (same k-ascending chain for
// (i,j) and (j,i), commutative norm sum), so only tiles with tn >= tm are computed; off-diagonal tiles
// are also written mirrored, transposed through LDS so that the mirrored rows leave as 128-byte pieces.
template <int EPI, bool SYM>
// (measured: k tile 32 -- half the barriers, two workgroups per CU -- 104 TF; forced to four waves per SIMD 112 TF;
// as is, three waves per SIMD, 115 TF on 20 000 x 80 000 x 768)
__global__ __launch_bounds__(256) void gemm_f32_exact_kernel(const float *__restrict__ A, const float *__restrict__ B,
                                                             int64_t M, int64_t N, int K,
                                                             const float *__restrict__ an,
                                                             const float *__restrict__ bn, float *__restrict__ C,
                                                             int64_t ldc, int tiles_m, int tiles_n, int vec_ok,
                                                             const unsigned *__restrict__ m_count) {
    __shared__ float Sm[2][2][XBK][XLD]; // one object: A stages, then B stages (33,792 B, reused by the mirror)
    float (&As)[2][XBK][XLD] = Sm[0];
    float (&Bs)[2][XBK][XLD] = Sm[1];

    int tm, tn;
    if (SYM) {
        // upper triangle enumerated column by column: id = tn*(tn+1)/2 + tm, tm <= tn
        const unsigned id = blockIdx.x;
        int c = (int)((__fsqrt_rn(8.0f * (float)id + 1.0f) - 1.0f) * 0.5f);
        while ((unsigned)(c + 1) * (unsigned)(c + 2) / 2u <= id) ++c;
        while ((unsigned)c * (unsigned)(c + 1) / 2u > id) --c;
        tn = c;
        tm = (int)(id - (unsigned)c * (unsigned)(c + 1) / 2u);
    } else {
        // m_count (only the first rows of a buffer sized for the worst case are live -- the re-ranking's fallback rows):
        // row-major tile ids, so that the live tiles (the first tile rows) are consecutive block ids = spread round-robin
        // over all XCDs.  The contiguous-chunk remap handed them all to the first XCD (7 ms instead of 1 for 687 x
        // 100 000), the grouped order without the remap still does when a single tile row is live (ids = 0 mod 8:
        // 0.38 ms instead of 0.11 for 112 x 20 000)
        if (m_count) {
            tm = (int)(blockIdx.x / (unsigned)tiles_n);
            tn = (int)(blockIdx.x % (unsigned)tiles_n);
        } else {
            tile_coords(xcd_remap(blockIdx.x, gridDim.x), tiles_m, tiles_n, 8, tm, tn);
        }
    }
    const int64_t m0 = (int64_t)tm * XBM, n0 = (int64_t)tn * XBN;
    if (m_count && m0 >= (int64_t)*m_count) return;   // rows past a device-side count (uniform per workgroup)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    // staging map: each thread moves XNR float4 of A and of B per K tile; idx -> (row, kc)
    constexpr int RSTEP = 256 / XKC;   // rows r0, r0 + RSTEP, ...
    const int r0 = tid / XKC, kc = tid % XKC;
    float4 ra[XNR], rb[XNR];
    const int nkt = (K + XBK - 1) / XBK;

    // fast staging (16-byte aligned rows, K a multiple of the k tile): unconditional 16-byte loads from row pointers
    // clamped to the last valid row (the padded rows' results are never stored); otherwise element-wise with bounds
    const bool fast = vec_ok != 0 && K % XBK == 0;
    const float *arow[XNR], *brow[XNR];
#pragma unroll
    for (int r = 0; r < XNR; ++r) {
        const int64_t ar = m0 + r0 + RSTEP * r, br = n0 + r0 + RSTEP * r;
        arow[r] = A + (ar < M ? ar : M - 1) * (int64_t)K + kc * 4;
        brow[r] = B + (br < N ? br : N - 1) * (int64_t)K + kc * 4;
    }
    auto gload = [&](int kt) {
        if (fast) {
#pragma unroll
            for (int r = 0; r < XNR; ++r) {
                ra[r] = *reinterpret_cast<const float4 *>(arow[r] + kt * XBK);
                rb[r] = *reinterpret_cast<const float4 *>(brow[r] + kt * XBK);
            }
            return;
        }
        const int k = kt * XBK + kc * 4;
#pragma unroll
        for (int r = 0; r < XNR; ++r) {
            ra[r] = load_k4(A, m0 + r0 + RSTEP * r, M, k, K, vec_ok != 0);
            rb[r] = load_k4(B, n0 + r0 + RSTEP * r, N, k, K, vec_ok != 0);
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int r = 0; r < XNR; ++r) {
            const int row = r0 + RSTEP * r;
            As[buf][kc * 4 + 0][row] = ra[r].x;
            As[buf][kc * 4 + 1][row] = ra[r].y;
            As[buf][kc * 4 + 2][row] = ra[r].z;
            As[buf][kc * 4 + 3][row] = ra[r].w;
            Bs[buf][kc * 4 + 0][row] = rb[r].x;
            Bs[buf][kc * 4 + 1][row] = rb[r].y;
            Bs[buf][kc * 4 + 2][row] = rb[r].z;
            Bs[buf][kc * 4 + 3][row] = rb[r].w;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    gload(0);
    sstore(0);
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) gload(kt + 1);
        // lane l holds A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31]; k ascends across MFMAs.  The operands of k pair
        // kp + 1 are read from LDS before the MFMAs of pair kp are issued (hipcc otherwise re-uses the operand registers
        // and exposes one LDS round trip per four MFMAs)
        float a0 = As[buf][lh][wm * 64 + li], a1 = As[buf][lh][wm * 64 + 32 + li];
        float b0 = Bs[buf][lh][wn * 64 + li], b1 = Bs[buf][lh][wn * 64 + 32 + li];
#pragma unroll
        for (int kp = 0; kp < XBK / 2; ++kp) {
            float a0n = 0.f, a1n = 0.f, b0n = 0.f, b1n = 0.f;
            if (kp + 1 < XBK / 2) {
                a0n = As[buf][2 * kp + 2 + lh][wm * 64 + li];
                a1n = As[buf][2 * kp + 2 + lh][wm * 64 + 32 + li];
                b0n = Bs[buf][2 * kp + 2 + lh][wn * 64 + li];
                b1n = Bs[buf][2 * kp + 2 + lh][wn * 64 + 32 + li];
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            a0 = a0n; a1 = a1n; b0 = b0n; b1 = b1n;
        }
        if (kt + 1 < nkt) sstore(buf ^ 1);
        __syncthreads();
    }

    // epilogue.  C/D map of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int64_t col = n0 + wn * 64 + j * 32 + li;
        if (col >= N) continue;
        const float bnv = (EPI >= EPI_LIN && bn == nullptr) ? 0.0f : bn[col];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row >= M) continue;
                const float dot = acc[i][j][r];
                float v;
                if (EPI == EPI_EUCLID) {
                    v = fmaf(-2.0f, dot, an[row] + bnv);
                } else if (EPI == EPI_LIN) {
                    v = dot + bnv;
                } else if (EPI == EPI_LIN_GELU) {
                    const float t = dot + bnv;   // model/clip/model.py:159-161  x * sigmoid(1.702 x), IEEE divide, accurate exp
                    v = __fdiv_rn(t, 1.0f + expf(-1.702f * t));
                } else if (EPI == EPI_LIN_RES) {
                    v = C[row * ldc + col] + (dot + bnv);
                } else if (EPI == EPI_LIN_RELU) {
                    v = dot + bnv;
                    v = v < 0.0f ? 0.0f : v;
                } else if (EPI == EPI_LIN_RES_RELU) {
                    v = C[row * ldc + col] + (dot + bnv);
                    v = v < 0.0f ? 0.0f : v;
                } else {
                    float c = dot * __fdiv_rn(1.0f, an[row] * bnv);
                    const float lo = (float)(-1.0 + 0.00001), hi = (float)(1.0 - 0.00001);
                    c = c < lo ? lo : (c > hi ? hi : c);
                    v = acosf(c);
                }
                C[row * ldc + col] = v;
                if (SYM) acc[i][j][r] = v; // keep the finished value for the mirrored write
            }
        }
    }
    if (SYM && tm != tn) {
        // mirrored tile: C[col][row] = C[row][col].  Each wave transposes its 64 x 64 block in two halves of
        // 32 rows through a private [64 cols][33] LDS patch (the staging buffers are free now).
        __syncthreads();
        static_assert(sizeof(Sm) >= 4 * 64 * 33 * sizeof(float), "mirror patches must fit the staging buffers");
        float *patch = &Sm[0][0][0][0] + wave * (64 * 33); // 4 x 8448 B
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    patch[(j * 32 + li) * 33 + (r & 3) + 8 * (r >> 2) + 4 * lh] = acc[i][j][r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // lanes 0..31 -> 32 consecutive original rows (= mirrored columns), lanes 32..63 the next column
            const int64_t mrow0 = m0 + wm * 64 + i * 32;
#pragma unroll 4
            for (int cc = 0; cc < 64; cc += 2) {
                const int cidx = cc + lh;
                const int64_t col = n0 + wn * 64 + cidx; // original column = mirrored row
                const int64_t row = mrow0 + li;          // original row = mirrored column
                if (col < N && row < M) C[col * ldc + row] = patch[cidx * 33 + li];
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host entry points
// ---------------------------------------------------------------------------------------------
extern "C" int mpreid_sqnorm_f32(const float *x, int64_t n, int d, float *out, mpreid_stream_t stream) {
    ARG_CHECK(x && out && n >= 0 && d > 0);
    if (n == 0) return MPREID_OK;
    hipLaunchKernelGGL(sqnorm_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, n, d, out,
                       0);
    LAUNCH_CHECK();
    return MPREID_OK;
}

extern "C" int mpreid_l2_normalize_f32(const float *x, int64_t n, int d, float eps, float *out,
                                       mpreid_stream_t stream) {
    ARG_CHECK(x && out && n >= 0 && d > 0);
    if (n == 0) return MPREID_OK;
    hipLaunchKernelGGL(l2_normalize_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, n, d,
                       eps, out);
    LAUNCH_CHECK();
    return MPREID_OK;
}

// implemented in gemm_f16.hip: fp16 one-pass distance (mode MPREID_GEMM_F16_FAST)
int mpreid_distance_f16_fast(const float *q, const float *g, int64_t nq, int64_t ng, int d, const float *qn,
                             const float *gn, float *out, int64_t ldo, int epi, void *ws, size_t ws_bytes,
                             hipStream_t stream);
size_t mpreid_distance_f16_ws_bytes(int64_t nq, int64_t ng, int d);
// 3-term fp16 split (mode MPREID_GEMM_F16_SPLIT3)
int mpreid_distance_f16_split3(const float *q, const float *g, int64_t nq, int64_t ng, int d, const float *qn,
                               const float *gn, float *out, int64_t ldo, int epi, void *ws, size_t ws_bytes,
                               hipStream_t stream);
size_t mpreid_distance_split3_ws_bytes(int64_t nq, int64_t ng, int d);

extern "C" size_t mpreid_distance_workspace_bytes(int64_t nq, int64_t ng, int d, int mode) {
    size_t b = align_up((size_t)(nq + ng) * sizeof(float), 256);
    if (mode == MPREID_GEMM_F16_FAST) b += mpreid_distance_f16_ws_bytes(nq, ng, d);
    if (mode == MPREID_GEMM_F16_SPLIT3) b += mpreid_distance_split3_ws_bytes(nq, ng, d);
    return b;
}

int mpreid_distance_launch(const float *q, const float *g, int64_t nq, int64_t ng, int d, const float *qn,
                           const float *gn, float *out, int64_t ldo, int epi, hipStream_t stream,
                           const unsigned *m_count) {
    const int tiles_m = (int)((nq + XBM - 1) / XBM), tiles_n = (int)((ng + XBN - 1) / XBN);
    const int vec_ok = (d % 4 == 0) && (((uintptr_t)q | (uintptr_t)g) % 16 == 0);
    const dim3 grid((unsigned)tiles_m * (unsigned)tiles_n);
    if (epi == EPI_EUCLID && q == g && nq == ng && qn == gn) {
        // all-pairs distance of one set: upper-triangular tiles + mirrored writes
        const dim3 tri((unsigned)tiles_m * (unsigned)(tiles_m + 1) / 2u);
        hipLaunchKernelGGL((gemm_f32_exact_kernel<EPI_EUCLID, true>), tri, dim3(256), 0, stream, q, g, nq, ng, d, qn, gn,
                           out, ldo, tiles_m, tiles_n, vec_ok, m_count);
    } else if (epi == EPI_EUCLID)
        hipLaunchKernelGGL((gemm_f32_exact_kernel<EPI_EUCLID, false>), grid, dim3(256), 0, stream, q, g, nq, ng, d, qn,
                           gn, out, ldo, tiles_m, tiles_n, vec_ok, m_count);
    else
        hipLaunchKernelGGL((gemm_f32_exact_kernel<EPI_COSINE, false>), grid, dim3(256), 0, stream, q, g, nq, ng, d, qn,
                           gn, out, ldo, tiles_m, tiles_n, vec_ok, m_count);
    LAUNCH_CHECK();
    return MPREID_OK;
}

// C[M][N] (row stride ldc) = A[M][K] x W[N][K]^T (+ bias[N]) with a linear-layer epilogue (EPI_LIN*); exact fp32 MFMA
// (k-ascending fmaf chains), any M, N, K: the GEMM of the all-fp32 encoder mode
int mpreid_gemm_f32_linear(const float *A, const float *Wt, int64_t M, int64_t N, int K, const float *bias, float *C,
                           int64_t ldc, int epi, hipStream_t stream) {
    const int tiles_m = (int)((M + XBM - 1) / XBM), tiles_n = (int)((N + XBN - 1) / XBN);
    const int vec_ok = (K % 4 == 0) && (((uintptr_t)A | (uintptr_t)Wt) % 16 == 0);
    const dim3 grid((unsigned)tiles_m * (unsigned)tiles_n);
    const float *none = nullptr;
    const unsigned *nocount = nullptr;
    switch (epi) {
    case EPI_LIN:
        hipLaunchKernelGGL((gemm_f32_exact_kernel<EPI_LIN, false>), grid, dim3(256), 0, stream, A, Wt, M, N, K, none, bias, C,
                           ldc, tiles_m, tiles_n, vec_ok, nocount);
        break;
    case EPI_LIN_GELU:
        hipLaunchKernelGGL((gemm_f32_exact_kernel<EPI_LIN_GELU, false>), grid, dim3(256), 0, stream, A, Wt, M, N, K, none, bias,
                           C, ldc, tiles_m, tiles_n, vec_ok, nocount);
        break;
    case EPI_LIN_RES:
        hipLaunchKernelGGL((gemm_f32_exact_kernel<EPI_LIN_RES, false>), grid, dim3(256), 0, stream, A, Wt, M, N, K, none, bias,
                           C, ldc, tiles_m, tiles_n, vec_ok, nocount);
        break;
    case EPI_LIN_RELU:
        hipLaunchKernelGGL((gemm_f32_exact_kernel<EPI_LIN_RELU, false>), grid, dim3(256), 0, stream, A, Wt, M, N, K, none, bias,
                           C, ldc, tiles_m, tiles_n, vec_ok, nocount);
        break;
    case EPI_LIN_RES_RELU:
        hipLaunchKernelGGL((gemm_f32_exact_kernel<EPI_LIN_RES_RELU, false>), grid, dim3(256), 0, stream, A, Wt, M, N, K, none,
                           bias, C, ldc, tiles_m, tiles_n, vec_ok, nocount);
        break;
    default:
        mpreid_set_error("gemm_f32_linear: unknown epilogue %d", epi);
        return MPREID_ERR_ARG;
    }
    LAUNCH_CHECK();
    return MPREID_OK;
}

static int distance_common(const float *q, const float *g, int64_t nq, int64_t ng, int d, float *out, int64_t ldo,
                           int mode, void *ws, size_t ws_bytes, mpreid_stream_t stream_, int epi) {
    ARG_CHECK(q && g && out && nq >= 0 && ng >= 0 && d > 0 && ldo >= ng);
    ARG_CHECK(mode == MPREID_GEMM_F32_EXACT || mode == MPREID_GEMM_F16_FAST || mode == MPREID_GEMM_F16_SPLIT3);
    if (nq == 0 || ng == 0) return MPREID_OK;
    if (ws == nullptr || ws_bytes < mpreid_distance_workspace_bytes(nq, ng, d, mode)) {
        mpreid_set_error("distance workspace too small: %zu < %zu", ws_bytes,
                         mpreid_distance_workspace_bytes(nq, ng, d, mode));
        return MPREID_ERR_WORKSPACE;
    }
    hipStream_t stream = (hipStream_t)stream_;
    // all-pairs distances of ONE set (same pointer): one norm vector serves rows and columns (mpreid_distance_launch and the
    // fp16 kernels recognise the case by the same test and compute the tiles on or above the diagonal only)
    const bool same = (q == g && nq == ng);
    float *qn = (float *)ws, *gn = same ? qn : qn + nq;
    const int nmode = (epi == EPI_COSINE) ? 1 : 0;
    hipLaunchKernelGGL(sqnorm_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, stream, q, nq, d, qn, nmode);
    if (!same) hipLaunchKernelGGL(sqnorm_kernel, dim3((unsigned)((ng + 3) / 4)), dim3(256), 0, stream, g, ng, d, gn, nmode);
    LAUNCH_CHECK();
    if (mode == MPREID_GEMM_F16_FAST) {
        char *rest = (char *)ws + align_up((size_t)(nq + ng) * sizeof(float), 256);
        return mpreid_distance_f16_fast(q, g, nq, ng, d, qn, gn, out, ldo, epi, rest,
                                        ws_bytes - (size_t)(rest - (char *)ws), stream);
    }
    if (mode == MPREID_GEMM_F16_SPLIT3) {
        char *rest = (char *)ws + align_up((size_t)(nq + ng) * sizeof(float), 256);
        return mpreid_distance_f16_split3(q, g, nq, ng, d, qn, gn, out, ldo, epi, rest,
                                          ws_bytes - (size_t)(rest - (char *)ws), stream);
    }
    return mpreid_distance_launch(q, g, nq, ng, d, qn, gn, out, ldo, epi, stream, nullptr);
}

extern "C" int mpreid_euclidean_distance_f32(const float *q, const float *g, int64_t nq, int64_t ng, int d, float *out,
                                             int64_t ldo, int mode, void *ws, size_t ws_bytes,
                                             mpreid_stream_t stream) {
    return distance_common(q, g, nq, ng, d, out, ldo, mode, ws, ws_bytes, stream, EPI_EUCLID);
}

extern "C" int mpreid_cosine_similarity_f32(const float *q, const float *g, int64_t nq, int64_t ng, int d, float *out,
                                            int64_t ldo, int mode, void *ws, size_t ws_bytes,
                                            mpreid_stream_t stream) {
    return distance_common(q, g, nq, ng, d, out, ldo, mode, ws, ws_bytes, stream, EPI_COSINE);
}
