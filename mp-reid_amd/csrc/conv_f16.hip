// conv_f16.hip — convolution as an implicit MFMA GEMM over NHWC fp16 activations (gfx950).
//
// The CLIP "RN50" tower (model/clip/model.py:10-148) is 1x1 and 3x3 stride-1 convolutions (strides are done by
// AvgPool2d), each followed by an eval-mode BatchNorm and mostly a ReLU.  With channels innermost (NHWC):
//
//   out[m][n] = act( bias[n] + sum_{tap, c} in[pixel(m) + offset(tap)][c] * w[n][tap][c]  (+ identity[m][n]) )
//
// i.e. C = A x W^T with M = B*H*W pixels, N = Cout, K = TAPS*Cin, where row m of the A operand for the K-step
// (tap, 64 channels) is 128 CONTIGUOUS bytes of the input image, shifted by the tap -- exactly the unit the LDS-DMA
// loader moves.  Nothing is materialised: no im2col buffer (it would cost ~9x the activation bytes, twice).
// BatchNorm is folded into w and bias on the host once (mpreid/ops.py); padding pixels read a 128-byte page of
// zeros; rows past M and channels past N are masked in the epilogue.
//
//   tile      128 (pixels) x 128 (channels) per 256-thread workgroup, 2x2 waves, 4x4 MFMA 16x16x32 f16 per wave;
//             128 x 64 (4x1 waves) for layers with <= 64 output channels
//   K step    64 halfs = one tap x 64 channels; A and W tiles double-buffered in LDS (64 KB -> 2 workgroups / CU)
//   staging   global_load_lds_dwordx4, 8 rows x 128 B per wave instruction; bank swizzle (chunk ^= row & 7) on the
//             SOURCE address and on the ds_read_b128 address (cdna_hip_programming.md §5.4 rule 21)
//   epilogue  fp32 accumulators -> wave-private LDS patch (two 32-row halves) -> rows of 8 channels per lane:
//             + bias (+ fp16 identity, 16-byte coalesced loads), ReLU, one rounding to fp16, 16-byte stores
//
// Requirements: Cin % 64 == 0 (the 32-channel stem tensors are stored with 64 channels, upper half zero),
// weights [Npad][TAPS*Cin] with Npad % 128 == 0 (zero rows), out / identity row stride ldo >= N, ldo % 8 == 0.
#include "common.h"
#include "conv_f16.h"

namespace {

typedef __attribute__((address_space(3))) void lds_ptr_t;
typedef const __attribute__((address_space(1))) void glb_ptr_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(const _Float16 *gsrc, unsigned char *lds_dst_wave_base) {
    __builtin_amdgcn_global_load_lds((glb_ptr_t *)gsrc, (lds_ptr_t *)lds_dst_wave_base, 16, 0, 0);
}

constexpr int CBM = 128, CBK = 64;
constexpr int C_TILE_BYTES = CBM * CBK * 2;      // 16 KB: the A (pixel) tile of a stage
constexpr int conv_stage_bytes(int BN) { return C_TILE_BYTES + BN * CBK * 2; }   // A + W
constexpr int conv_lds_bytes(int BN) { return 2 * conv_stage_bytes(BN); }        // 64 KB (BN 128) / 48 KB (BN 64)

// BN = output channels per tile: 128 (2 x 2 waves of 64 x 64) or 64 (4 x 1 waves of 32 x 64) for the layers with
// <= 64 output channels (stem, layer1): a 128-wide tile would spend half of its MFMAs on zero weight rows
template <int TAPS, bool RELU, int BN, bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(ConvArgs g, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int C_STAGE_BYTES = conv_stage_bytes(BN);
    constexpr int WM = (BN == 128) ? 64 : 32;   // pixel rows per wave
    constexpr int MI = WM / 16;                 // 16-row MFMA tiles per wave
    constexpr int BROWS = BN / 4;               // weight rows staged per wave
    int tm, tn;
    tile_coords(xcd_remap(blockIdx.x, gridDim.x), tiles_m, tiles_n, 8, tm, tn);
    const int m0 = tm * CBM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (BN == 128) ? (wave >> 1) : wave, wn = (BN == 128) ? (wave & 1) : 0;
    const int C = g.C;
    const int cpt = C / CBK;                       // 64-channel blocks per tap
    const int AS = SPLIT ? 2 * C : C;              // row stride of the activations in halfs (pairs: hi | lo)
    const int K = TAPS * C * (SPLIT ? 2 : 1);      // row length of the weights (pairs: per (tap, 64 channels) a hi' slab and a lo' slab)

    // ---- staging: wave w moves rows [32w, 32w+32) of both tiles, 8 rows per DMA instruction ----
    const int srow = lane >> 3;
    const int gchunk = (lane & 7) ^ srow;
    const _Float16 *a_row[4];
    unsigned tapmask[4]; // bit t: tap t of this row reads a real pixel
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int m = m0 + wave * 32 + t * 8 + srow;
        a_row[t] = g.act + (int64_t)m * AS + gchunk * 8;
        unsigned mask = 0;
        if (m < g.M) {
            if (TAPS == 1) {
                mask = 1;
            } else {
                const int x = m % g.W, y = (m / g.W) % g.H;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
                    if (yy >= 0 && yy < g.H && xx >= 0 && xx < g.W) mask |= 1u << tap;
                }
            }
        }
        tapmask[t] = mask;
    }
    const _Float16 *zero = g.zero_page + gchunk * 8;
    const _Float16 *b_src = g.wgt + (int64_t)(n0 + wave * BROWS + srow) * K + gchunk * 8;
    auto stage = [&](int s, int kt) {
        unsigned char *abase = smem + s * C_STAGE_BYTES + wave * 4096;
        unsigned char *bbase = smem + s * C_STAGE_BYTES + C_TILE_BYTES + wave * (BROWS * 128);
        const int tap = (TAPS == 1) ? 0 : kt / cpt;
        const int c0 = (kt - tap * cpt) * CBK;
        const int64_t shift = (TAPS == 1) ? (int64_t)c0 : (int64_t)((tap / 3 - 1) * g.W + (tap % 3 - 1)) * AS + c0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const _Float16 *src = ((tapmask[t] >> tap) & 1u) ? a_row[t] + shift : zero;
            dma16(src, abase + t * 1024);
            if (t < BROWS / 8) dma16(b_src + (int64_t)t * 8 * K + kt * CBK, bbase + t * 1024);
        }
    };

    // ---- fragment addresses ----
    const int frow = lane & 15, fq = lane >> 4;
    const int a_row_off = (wm * WM + frow) * 128;
    const int b_row_off = (wn * 64 + frow) * 128;
    int ksw[2];
    ksw[0] = ((0 + fq) ^ (lane & 7)) << 4;
    ksw[1] = ((4 + fq) ^ (lane & 7)) << 4;

    f32x4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int s) {
        const unsigned char *at = smem + s * C_STAGE_BYTES;
        const unsigned char *bt = at + C_TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 af[MI], bf[4];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const f16x8 *>(at + a_row_off + i * 2048 + ksw[ks]);
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const f16x8 *>(bt + b_row_off + j * 2048 + ksw[ks]);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    };

    const int nkt = K / CBK;
    if constexpr (SPLIT) {
        // Pair form: ONE stage per block (tap, 64 channels) holding all four tiles -- A hi, A lo, W hi', W lo' (64 KB at
        // BN = 128) -- single-buffered, and the three products hi.hi', lo.hi', hi.lo' run from it (96 matrix instructions per
        // wave between two barriers; every fragment is read from LDS once).  The double-buffered two-tile steps of the fp16
        // form staged six tiles per block and paid one DMA round trip (~1.5 us) per 32 matrix instructions (0.3 us): 1.7 us
        // per step measured, 0.62 PF executed.  Here a workgroup still waits for its stage, but a stage carries three times
        // the work and the CU's second workgroup computes meanwhile.
        const int nblk = nkt / 2;
        unsigned char *a_hi = smem, *a_lo = smem + C_TILE_BYTES, *w_hi = smem + 2 * C_TILE_BYTES, *w_lo = w_hi + BN * 128;
        for (int kb = 0; kb < nblk; ++kb) {
            const int tap = (TAPS == 1) ? 0 : kb / cpt;
            const int c0 = (kb - tap * cpt) * CBK;
            const int64_t shift = (TAPS == 1) ? (int64_t)c0 : (int64_t)((tap / 3 - 1) * g.W + (tap % 3 - 1)) * AS + c0;
            __syncthreads();   // every wave is done with the previous block's tiles
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bool real = (tapmask[t] >> tap) & 1u;
                dma16(real ? a_row[t] + shift : zero, a_hi + wave * 4096 + t * 1024);
                dma16(real ? a_row[t] + shift + C : zero, a_lo + wave * 4096 + t * 1024);
                if (t < BROWS / 8) {
                    dma16(b_src + (int64_t)t * 8 * K + (2 * kb) * CBK, w_hi + wave * (BROWS * 128) + t * 1024);
                    dma16(b_src + (int64_t)t * 8 * K + (2 * kb + 1) * CBK, w_lo + wave * (BROWS * 128) + t * 1024);
                }
            }
            __syncthreads();   // (waits for the DMA: vmcnt(0), then the barrier)
            f16x8 ah[2][MI], al[2][MI], bh[2][4], bl[2][4];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    ah[ks][i] = *reinterpret_cast<const f16x8 *>(a_hi + a_row_off + i * 2048 + ksw[ks]);
                    al[ks][i] = *reinterpret_cast<const f16x8 *>(a_lo + a_row_off + i * 2048 + ksw[ks]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bh[ks][j] = *reinterpret_cast<const f16x8 *>(w_hi + b_row_off + j * 2048 + ksw[ks]);
                    bl[ks][j] = *reinterpret_cast<const f16x8 *>(w_lo + b_row_off + j * 2048 + ksw[ks]);
                }
            }
            // (the order of the two-tile version: product by product, the two 32-wide k steps inside)
#pragma unroll
            for (int prod = 0; prod < 3; ++prod)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(prod == 1 ? al[ks][i] : ah[ks][i],
                                                                               prod == 2 ? bl[ks][j] : bh[ks][j], acc[i][j], 0, 0, 0);
        }
    } else {
        stage(0, 0);
        __syncthreads();
        int cur = 0;
        for (int kt = 0; kt < nkt - 1; ++kt) {
            stage(cur ^ 1, kt + 1);
            compute(cur);
            __syncthreads();
            cur ^= 1;
        }
        compute(cur);
    }

    // ---- epilogue: C/D map of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg ----
    __syncthreads();
    float *wreg = reinterpret_cast<float *>(smem) + wave * (32 * 68);
    const int ch = lane & 7;                       // 8-channel chunk of the wave's 64 columns
    const int n = n0 + wn * 64 + ch * 8;
    float bias[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias[e] = g.bias[n + e]; // bias is padded to Npad
#pragma unroll
    for (int half = 0; half < WM / 32; ++half) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    wreg[(ii * 16 + fq * 4 + r) * 68 + j * 16 + frow] = acc[half * 2 + ii][j][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int lr = it * 8 + (lane >> 3);
            const int m = m0 + wm * WM + half * 32 + lr;
            const float4 v0 = *reinterpret_cast<const float4 *>(wreg + lr * 68 + ch * 8);
            const float4 v1 = *reinterpret_cast<const float4 *>(wreg + lr * 68 + ch * 8 + 4);
            float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            if constexpr (SPLIT) {
                if (g.out_pairs) {   // the consumer is a pair GEMM: its operand directly (ReLU applied), no fp32 round trip
                    if (m < g.M && n < g.pair_c) {
                        f16x8 hi, lo;
                        float yv[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float y = fmaf(v[e], g.oscale, bias[e]);
                            yv[e] = (n + e < g.N && y > 0.f) ? y : 0.f;
                        }
                        split_pairs<8>(yv, hi, lo);
                        _Float16 *o = g.out_pairs + (int64_t)m * 2 * g.pair_c + n;
                        *reinterpret_cast<f16x8 *>(o) = hi;
                        *reinterpret_cast<f16x8 *>(o + g.pair_c) = lo;
                    }
                } else if (m < g.M && n < g.Npad) {
                    float *o = g.out32 + (int64_t)m * g.ldo + n;
                    *reinterpret_cast<float4 *>(o) = make_float4(fmaf(v[0], g.oscale, bias[0]), fmaf(v[1], g.oscale, bias[1]),
                                                                 fmaf(v[2], g.oscale, bias[2]), fmaf(v[3], g.oscale, bias[3]));
                    *reinterpret_cast<float4 *>(o + 4) = make_float4(fmaf(v[4], g.oscale, bias[4]), fmaf(v[5], g.oscale, bias[5]),
                                                                     fmaf(v[6], g.oscale, bias[6]), fmaf(v[7], g.oscale, bias[7]));
                }
            } else if (m < g.M && n < g.N) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] + bias[e];
                if (g.identity) {
                    const f16x8 idv = *reinterpret_cast<const f16x8 *>(g.identity + (int64_t)m * g.ldo + n);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = v[e] + (float)idv[e];
                }
                f16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (_Float16)((RELU && v[e] < 0.f) ? 0.f : v[e]);
                *reinterpret_cast<f16x8 *>(g.out + (int64_t)m * g.ldo + n) = o;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

} // namespace

template <int TAPS, bool RELU, int BN, bool SPLIT = false>
static int launch_conv_variant(const ConvArgs &a, hipStream_t stream) {
    const int tiles_m = (a.M + CBM - 1) / CBM, tiles_n = (a.N + BN - 1) / BN;
    static PerDeviceOnce attr_once;
    {
        const int rc = attr_once.run([]() -> int {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_gemm_kernel<TAPS, RELU, BN, SPLIT>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, conv_lds_bytes(BN)));
            return MPREID_OK;
        });
        if (rc) return rc;
    }
    hipLaunchKernelGGL((conv_gemm_kernel<TAPS, RELU, BN, SPLIT>), dim3((unsigned)(tiles_m * tiles_n)), dim3(256), conv_lds_bytes(BN),
                       stream, a, tiles_m, tiles_n);
    LAUNCH_CHECK();
    return 0;
}

int launch_conv_f16(const ConvArgs &a, hipStream_t stream) {
    if (a.split) {   // the pair form: 3x3 only, fp32 out
        ARG_CHECK(a.act && a.wgt && a.bias && (a.out32 || a.out_pairs) && a.zero_page && a.taps == 9 && a.C > 0 && a.C % CBK == 0 && a.M > 0);
        ARG_CHECK(!a.out_pairs || (a.pair_c % 64 == 0 && a.pair_c >= a.N));
        ARG_CHECK(a.Npad % 128 == 0 && a.N > 0 && a.Npad >= a.N && a.ldo % 4 == 0 && a.ldo >= a.Npad && a.H > 0 && a.W > 0 &&
                  a.M % (a.H * a.W) == 0);
        return a.N <= 64 ? launch_conv_variant<9, false, 64, true>(a, stream) : launch_conv_variant<9, false, 128, true>(a, stream);
    }
    ARG_CHECK(a.act && a.wgt && a.bias && a.out && a.zero_page);
    ARG_CHECK((a.taps == 1 || a.taps == 9) && a.C > 0 && a.C % CBK == 0 && a.M > 0 && a.N > 0);
    ARG_CHECK(a.Npad % 128 == 0 && a.Npad >= a.N && a.N % 8 == 0 && a.ldo % 8 == 0 && a.ldo >= a.N);
    ARG_CHECK(a.taps == 1 || (a.H > 0 && a.W > 0 && a.M % (a.H * a.W) == 0));
    const bool narrow = a.N <= 64;   // 64-channel tiles: no MFMAs on the zero rows that pad the weights to 128
    if (a.taps == 1) {
        if (a.relu) return narrow ? launch_conv_variant<1, true, 64>(a, stream) : launch_conv_variant<1, true, 128>(a, stream);
        return narrow ? launch_conv_variant<1, false, 64>(a, stream) : launch_conv_variant<1, false, 128>(a, stream);
    }
    if (a.relu) return narrow ? launch_conv_variant<9, true, 64>(a, stream) : launch_conv_variant<9, true, 128>(a, stream);
    return narrow ? launch_conv_variant<9, false, 64>(a, stream) : launch_conv_variant<9, false, 128>(a, stream);
}

// unit-test / micro-benchmark entry: one convolution layer on NHWC fp16 (include/mpreid.h)
extern "C" int mpreid_conv_f16_nhwc(const void *act, int batch, int h, int w, int cin, const void *wgt, const float *bias,
                                    int cout, int cout_pad, int taps, const void *identity, int relu, void *out,
                                    const void *zero_page, mpreid_stream_t stream) {
    ConvArgs a{};
    a.act = (const _Float16 *)act;
    a.wgt = (const _Float16 *)wgt;
    a.bias = bias;
    a.identity = (const _Float16 *)identity;
    a.out = (_Float16 *)out;
    a.zero_page = (const _Float16 *)zero_page;
    a.H = h;
    a.W = w;
    a.C = cin;
    a.M = batch * h * w;
    a.N = cout;
    a.Npad = cout_pad;
    a.ldo = cout;
    a.taps = taps;
    a.relu = relu;
    return launch_conv_f16(a, (hipStream_t)stream);
}
