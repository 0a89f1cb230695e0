// vit.hip — CLIP ViT-B/16 image encoder forward for gfx950.
//
// Reference: model/clip/model.py:415-479 (VisionTransformer.forward), :278-281
// (ResidualAttentionBlock), :150-156 (LayerNorm, fp32 statistics), :159-161 (QuickGELU);
// model/make_model.py:81-115 (CLS of x12 and of x12 @ proj, eval BatchNorm necks, concat).
//
// Precision plan: residual stream, LayerNorm statistics, softmax and all accumulators fp32;
// GEMM operands (LN outputs, Q/K/V, attention output, MLP hidden, weights) fp16 -> MFMA.
//
// HBM layout (workspace, M = B*L tokens, padded to 128 rows):
//   patches fp16 [B*P pad][3*p*p]   im2col of the image, inner order (c, kh, kw)
//   x       fp32 [M pad][W]         residual stream
//   a       fp16 [M pad][W]         LN output, then attention output (GEMM A operands)
//   qkv     fp16 [M pad][3W]        in_proj output, q | k | v column blocks, head h = cols [64h, 64h+64)
//   hbuf    fp16 [M pad][4W]        MLP hidden after QuickGELU
// Kernels: im2col (HBM), GEMM+epilogues (MFMA, gemm_f16.hip), LayerNorm (HBM), attention (MFMA +
// LDS; L = 129 keys fit one workgroup), head (L2).
#include <cstdlib>
#include <mutex>

#include "gemm_f16.h"

// ---------------------------------------------------------------------------------------------
// im2col: img [B][3][H][W] fp32 -> patches [MPpad][3*p*p] fp16; rows >= B*P are zero
//
// VIEW = the test-time-augmentation views of processor/processor_uniprompt_stage2.py:605-633, applied while
// gathering (the reference materialises each view as a new [B,3,H,W] tensor and runs the model on it):
//   0 original   1 torch.flip(img, [3]) (mirror in W)   2 pseudo-IR: img.mean(dim=1) in all 3 channels
//   3 pseudo-RGB: channel 0 in all 3 channels
// ---------------------------------------------------------------------------------------------
// SPLIT (the `split` precision mode): rows are fp16 pairs [hi(Kp) | lo(Kp)] with hi + lo = the fp32 value to ~2^-22
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
template <bool SPLIT>
__device__ __forceinline__ void store_patch_row(_Float16 *__restrict__ out, int64_t m, int Kp, int ch, const float (&v)[8]) {
    h8_t hi, lo;
    split_pairs<8>(v, hi, lo);
    if constexpr (SPLIT) {
        *reinterpret_cast<h8_t *>(out + m * 2 * Kp + ch * 8) = hi;
        *reinterpret_cast<h8_t *>(out + m * 2 * Kp + Kp + ch * 8) = lo;
    } else {
        *reinterpret_cast<h8_t *>(out + m * Kp + ch * 8) = hi;
    }
}

template <int VIEW, bool SPLIT>
__global__ __launch_bounds__(256) void im2col_kernel(const float *__restrict__ img, int B, int H, int Wd, int p,
                                                     int stride, int h_res, int w_res, _Float16 *__restrict__ out,
                                                     int mp_pad) {
    const int Kp = 3 * p * p;
    const int chunks = Kp / 8; // 8 consecutive kw per thread
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (int64_t)mp_pad * chunks) return;
    const int m = (int)(gid / chunks), ch = (int)(gid % chunks);
    const int P = h_res * w_res;
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (m < B * P) {
        const int b = m / P, pi = m % P;
        const int ph = pi / w_res, pw = pi % w_res;
        const int k = ch * 8;
        const int c = k / (p * p), kh = (k % (p * p)) / p, kw = k % p;
        const int64_t plane = (int64_t)H * Wd;
        const int csrc = (VIEW == 2 || VIEW == 3) ? 0 : c;
        const float *row = img + ((int64_t)b * 3 + csrc) * plane + (int64_t)(ph * stride + kh) * Wd;
        const int x0 = pw * stride + kw;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int x = (VIEW == 1) ? (Wd - 1 - (x0 + j)) : (x0 + j);
            float v = row[x];
            // torch mean over the channel dim: ((c0 + c1) + c2) / 3
            if (VIEW == 2) v = __fdiv_rn((v + row[plane + x]) + row[2 * plane + x], 3.0f);
            o[j] = v;
        }
    }
    store_patch_row<SPLIT>(out, m, Kp, ch, o);
}

// uint8 variant: img [B][H][W][3] (HWC, what PIL / cv2 hand over after Resize).  ToTensor (x / 255) and
// Normalize ((x - mean) / std) of the reference's val_transforms (datasets/make_dataloader.py:57-61) are applied
// while gathering, in fp32, before the fp16 rounding -- the same values the fp32 entry point receives.  The views
// act on the normalised values, as in the reference.
template <int VIEW, bool SPLIT>
__global__ __launch_bounds__(256) void im2col_u8_kernel(const unsigned char *__restrict__ img, int B, int H, int Wd, int p,
                                                        int stride, int h_res, int w_res, float m0, float m1, float m2,
                                                        float s0, float s1, float s2, _Float16 *__restrict__ out,
                                                        int mp_pad) {
    const int Kp = 3 * p * p;
    const int chunks = Kp / 8;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (int64_t)mp_pad * chunks) return;
    const int m = (int)(gid / chunks), ch = (int)(gid % chunks);
    const int P = h_res * w_res;
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (m < B * P) {
        const int b = m / P, pi = m % P;
        const int ph = pi / w_res, pw = pi % w_res;
        const int k = ch * 8;
        const int c = k / (p * p), kh = (k % (p * p)) / p, kw = k % p;
        const int csrc = (VIEW == 2 || VIEW == 3) ? 0 : c;
        const float mean = csrc == 0 ? m0 : (csrc == 1 ? m1 : m2), sd = csrc == 0 ? s0 : (csrc == 1 ? s1 : s2);
        const unsigned char *row = img + ((int64_t)b * H + (ph * stride + kh)) * Wd * 3;
        const int x0 = pw * stride + kw;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int x = (VIEW == 1) ? (Wd - 1 - (x0 + j)) : (x0 + j);
            float v = __fdiv_rn(__fdiv_rn((float)row[x * 3 + csrc], 255.0f) - mean, sd);
            if (VIEW == 2) {
                const float v1 = __fdiv_rn(__fdiv_rn((float)row[x * 3 + 1], 255.0f) - m1, s1);
                const float v2 = __fdiv_rn(__fdiv_rn((float)row[x * 3 + 2], 255.0f) - m2, s2);
                v = __fdiv_rn((v + v1) + v2, 3.0f);
            }
            o[j] = v;
        }
    }
    store_patch_row<SPLIT>(out, m, Kp, ch, o);
}

// x[b*L + 0][:] = class_embedding + pos[0] (+ cv_emb[b])      model/clip/model.py:419-422
__global__ __launch_bounds__(256) void cls_token_kernel(const float *__restrict__ cls, const float *__restrict__ pos,
                                                        const float *__restrict__ cv, int B, int L, int W,
                                                        float *__restrict__ x) {
    const int b = blockIdx.x;
    for (int k = threadIdx.x; k < W; k += 256) {
        float v = cls[k];
        if (cv) v = v + cv[(int64_t)b * W + k];
        x[(int64_t)b * L * W + k] = v + pos[k];
    }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over the last dim (biased variance, eps 1e-5, fp32 stats); one wave per row.
// MODE 0: write fp32 (may alias the input: ln_pre is in place); 1: fp16 (GEMM operand); 2: fp16 pair [hi(W) | lo(W)]
// per row (GEMM operand of the `split` precision mode).
// ---------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void layernorm_kernel(const float *__restrict__ x, int64_t rows, int W,
                                                        const float *__restrict__ gamma,
                                                        const float *__restrict__ beta, void *__restrict__ out,
                                                        int64_t row_stride_in) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *xr = x + row * row_stride_in;
    float4 v[4]; // W <= 1024; statically unrolled so that v[] stays in registers
    bool act[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        act[i] = (i * 256 + lane * 4) < W;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (act[i]) {
            typedef float ln_f4 __attribute__((ext_vector_type(4)));
            const ln_f4 t = __builtin_nontemporal_load(reinterpret_cast<const ln_f4 *>(xr + i * 256 + lane * 4));
            v[i] = make_float4(t[0], t[1], t[2], t[3]);
        }
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    const float mean = s / (float)W;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (act[i]) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + b * b) + (c * c + d * d);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) q += __shfl_xor(q, off, 64);
    const float rstd = 1.0f / __fsqrt_rn(q / (float)W + 1e-5f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (!act[i]) continue;
        const int k = i * 256 + lane * 4;
        const float4 gm = *reinterpret_cast<const float4 *>(gamma + k);
        const float4 bt = *reinterpret_cast<const float4 *>(beta + k);
        float4 y;
        y.x = (v[i].x - mean) * rstd * gm.x + bt.x;
        y.y = (v[i].y - mean) * rstd * gm.y + bt.y;
        y.z = (v[i].z - mean) * rstd * gm.z + bt.z;
        y.w = (v[i].w - mean) * rstd * gm.w + bt.w;
        if (MODE == 1) {
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            h4 o = {(_Float16)y.x, (_Float16)y.y, (_Float16)y.z, (_Float16)y.w};
            *reinterpret_cast<h4 *>(reinterpret_cast<_Float16 *>(out) + row * (int64_t)W + k) = o;
        } else if (MODE == 2) {
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            h4 hi, lo;
            {
                const float yv[4] = {y.x, y.y, y.z, y.w};
                split_pairs<4>(yv, hi, lo);
            }
            _Float16 *orow = reinterpret_cast<_Float16 *>(out) + row * 2 * (int64_t)W;
            *reinterpret_cast<h4 *>(orow + k) = hi;
            *reinterpret_cast<h4 *>(orow + W + k) = lo;
        } else {
            *reinterpret_cast<float4 *>(reinterpret_cast<float *>(out) + row * (int64_t)W + k) = y;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// attention: softmax(q k^T / sqrt(64)) v per (image, head); L <= 16*KTP keys live in LDS.
//
// One workgroup per (image, head); each wave owns 16-query tiles.  Computed "key on the lane":
//   S^T = K Q^T   (A = K rows from LDS, B = Q fragment straight from global)  -> C: col = query
//   O^T = V^T P^T (A = V^T from LDS,   B = P^T = the S^T accumulators, converted in registers)
// so the softmax reduction over keys is in-lane + 2 shuffles and P never touches LDS
// (cdna_hip_programming.md §3 "An accumulator tile as the next MFMA's operand").
// ---------------------------------------------------------------------------------------------
typedef _Float16 h4_t __attribute__((ext_vector_type(4)));

// K / V prefetch registers are NAMED scalars (k0..k4, v0..v4) expanded by macros: as arrays they were kept
// in scratch memory by hipcc, doubling the memory traffic of the kernel.
// thread -> (row, 16-byte chunk), 8 lanes per 128-byte row, so every wave instruction covers 8 whole cache
// lines.  Rows past L are clamped (always a valid address): a guarded load would put every load in its own
// branch with a vmcnt(0) behind it; clamped copies are harmless because keys >= L are masked before softmax.
#define ATT_FOR_EACH_ITER(X) X(0) X(1) X(2) X(3) X(4)
// K / V are read exactly once and O is read only by a later kernel: streaming loads / stores keep them from evicting
// the operand panels of the GEMMs that run beside this kernel on other streams
typedef unsigned att_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 load_nt16(const _Float16 *p) {
    const att_u4 v = __builtin_nontemporal_load(reinterpret_cast<const att_u4 *>(p));
    return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void store_nt16(_Float16 *p, const uint4 &v) {
    const att_u4 nv = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(nv, reinterpret_cast<att_u4 *>(p));
}
#define ATT_DECL(i) uint4 k##i = make_uint4(0, 0, 0, 0), v##i = k##i;
#define ATT_LOAD(i)                                                                                  \
    if constexpr (K_ITERS > i) {                                                                     \
        const int idx_ = tid + i * NT;                                                               \
        const int row_ = idx_ >> 3, c_ = idx_ & 7;                                                   \
        const int rc_ = row_ < L ? row_ : L - 1;                                                     \
        k##i = load_nt16(pbase_ + (int64_t)rc_ * ld + W + c_ * 8);                                   \
        v##i = load_nt16(pbase_ + (int64_t)rc_ * ld + 2 * W + c_ * 8);                               \
    }
#define ATT_STORE(i)                                                                                 \
    if constexpr (K_ITERS > i) {                                                                     \
        const int idx_ = tid + i * NT;                                                               \
        const int row_ = idx_ >> 3, c_ = idx_ & 7;                                                   \
        if (idx_ < KEYS * 8) {                                                                       \
            *reinterpret_cast<uint4 *>(Ks + row_ * 128 + ((c_ ^ (row_ & 7)) << 4)) = k##i;           \
            *reinterpret_cast<uint4 *>(Vs + row_ * 128 + ((c_ ^ (((row_ >> 1) & 3) << 1)) << 4)) = v##i; \
        }                                                                                            \
    }
#define ATT_PREFETCH(pr)                                                                             \
    {                                                                                                \
        const int pb_ = (pr) / heads, ph_ = (pr) - pb_ * heads;                                      \
        const _Float16 *pbase_ = qkv + (int64_t)pb_ * L * ld + ph_ * 64;                             \
        ATT_FOR_EACH_ITER(ATT_LOAD)                                                                  \
    }

// All-reduce over the lanes l, l^16, l^32, l^48 (the four key groups of a query column) with gfx950's
// v_permlane16_swap / v_permlane32_swap: with both operands = x, the two results hold (own, partner) in one order or the
// other for every lane, so op(r[0], r[1]) is the xor-16 (then xor-32) exchange -- two VALU operations per step where
// __shfl_xor compiles to ds_bpermute_b32 (an LDS crossbar round trip, ~120 cycles; four per query tile).  Same bits: the
// operations are commutative.
__device__ __forceinline__ float xor16_32_max(float x) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor16_32_sum(float x) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

typedef short att_s4 __attribute__((__vector_size__(4 * sizeof(short))));
typedef __attribute__((address_space(3))) att_s4 att_lds_s4;

// Persistent form: a workgroup walks (image, head) pairs pair = blockIdx.x, + gridDim.x, ...  The
// K / V / Q data of the NEXT pair are requested into registers before the current pair is computed
// from LDS, so HBM loads stay in flight during the MFMA/softmax phase (the non-persistent form kept
// loads in flight only about half of the time: 2.5 TB/s).
template <int KTP, int NW, bool EXACT>
__global__ __launch_bounds__(64 * NW, (NW == 8 && KTP <= 10) ? 4 : 2) void attention_kernel(const _Float16 *__restrict__ qkv, int L, int W, int heads,
                                                               _Float16 *__restrict__ out, int q_tiles,
                                                               int total_pairs, int dbg_) {
#ifdef MPREID_ABLATION
    const int dbg = dbg_;
#else
    constexpr int dbg = 0;   // (see attention_split_kernel)
    (void)dbg_;
#endif
    constexpr int KEYS = KTP * 16;
    constexpr int NT = 64 * NW;
    constexpr int K_ITERS = (KEYS * 8 + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // K  [KEYS][128 B], 16-byte chunk ^= row & 7           (ds_read_b128 of A fragments, conflict free)
    // V  [KEYS][128 B], 16-byte chunk ^= ((row>>1)&3) << 1 (ds_read_b64_tr_b16 of 4-key x 16-d blocks, conflict free)
    unsigned char *Ks = smem;
    unsigned char *Vs = smem + KEYS * 128;
    constexpr int OS = 72;                                                // halfs per row of an O patch (144 B)
    _Float16 *Ot = reinterpret_cast<_Float16 *>(smem + 2 * KEYS * 128);   // [NW][16][OS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int64_t ld = 3 * (int64_t)W;
    const int nqt = (q_tiles > 0) ? q_tiles : (L + 15) / 16;
    const float scale_log2e = 0.125f * 1.44269504088896340736f;

    static_assert(K_ITERS <= 5, "extend ATT_FOR_EACH_ITER");
    ATT_FOR_EACH_ITER(ATT_DECL)

    int pair = blockIdx.x;
    if (pair >= total_pairs) return;
    if (!(dbg & 1)) ATT_PREFETCH(pair)
    for (; pair < total_pairs; pair += gridDim.x) {
        const int b = pair / heads, h = pair - b * heads;
        const _Float16 *base = qkv + (int64_t)b * L * ld + h * 64;
        __syncthreads(); // every wave is done reading the previous pair from LDS
        // ---- registers -> LDS (rows >= L hold a clamped, finite copy of row L-1: masked before the softmax) ----
        ATT_FOR_EACH_ITER(ATT_STORE)
        f16x8 qf[2];
        {   // Q fragment (B operand: B[k = d][col = query]) of this wave's first tile, requested before the
            // barrier; rows past L are clamped and only feed discarded output columns
            const int q = wave * 16 + fr;
            const int qc = q < L ? q : L - 1;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                qf[ks] = *reinterpret_cast<const f16x8 *>(base + (int64_t)qc * ld + ks * 32 + fq * 8);
        }
        __syncthreads();
        {   // unconditional (a clamped pair index on the last trip): keeps the register arrays out of scratch
            const int nxt = pair + (int)gridDim.x < total_pairs ? pair + (int)gridDim.x : pair;
            if (!(dbg & 1)) ATT_PREFETCH(nxt)
        }

        for (int qt = wave; qt < ((dbg & 2) ? 0 : nqt); qt += NW) {
            const int q = qt * 16 + fr;
            if (qt != wave) { // later tiles of this wave (only when there are more tiles than waves)
                const int qc = q < L ? q : L - 1;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    qf[ks] = *reinterpret_cast<const f16x8 *>(base + (int64_t)qc * ld + ks * 32 + fq * 8);
            }
        // S^T tiles, branch free.  EXACT: KTP is the even round-up of ceil(L/16), so tiles 0..KTP-3
        // hold only valid keys and only the last two can contain keys >= L (K rows past L are zero in
        // LDS; their scores are forced to -huge).  !EXACT (KTP larger than needed): mask every tile.
        f32x4 s[KTP];
        float mx = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < KTP; ++kt) {
            s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const f16x8 kf = *reinterpret_cast<const f16x8 *>(Ks + (kt * 16 + fr) * 128 +
                                                                   (((ks * 4 + fq) ^ (lane & 7)) << 4));
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[ks], s[kt], 0, 0, 0);
            }
            // keep at most two tiles' worth of K fragments in flight (bounds the register live ranges
            // so that two workgroups fit a CU)
            if (kt & 1) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int kt = 0; kt < KTP; ++kt) {
            if (!EXACT || kt >= KTP - 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kt * 16 + fq * 4 + r >= L) s[kt][r] = -3.0e38f;
            }
            mx = fmaxf(mx, fmaxf(fmaxf(s[kt][0], s[kt][1]), fmaxf(s[kt][2], s[kt][3])));
        }
        mx = xor16_32_max(mx);
        // p = exp2(s*c - mx*c): one fma + one v_exp per element; masked entries give exp2(-huge) = 0
        const float nmx = -mx * scale_log2e;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < KTP; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __builtin_amdgcn_exp2f(fmaf(s[kt][r], scale_log2e, nmx));
                s[kt][r] = p;
                sum += p;
            }
        sum = xor16_32_sum(sum);
        // O^T = V^T P^T over 32-key steps; P^T fragment element j <-> key 32*s2 + 16*(j>>2) + 4*fq + (j&3)
        // (V^T columns past L are zero in LDS and their P is 0, so every step runs unconditionally)
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s2 = 0; s2 < KTP / 2; ++s2) {
            f16x8 pf;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                pf[j] = (_Float16)s[2 * s2][j];
                pf[4 + j] = (_Float16)s[2 * s2 + 1][j];
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                // A operand = V^T: lane (d = dt*16 + fr, key group fq) needs keys 32*s2 + 4*fq + {0..3} and + 16.
                // One ds_read_b64_tr_b16 hands lane i of a 16-lane group column i of a 4-row x 16-column block;
                // lane 4q+p of the group supplies the address of row q, columns 4p..4p+3.
                const int trq = fr >> 2, trp = fr & 3;
                const int row0 = s2 * 32 + fq * 4 + trq;
                const int chunk = dt * 2 + (trp >> 1);
                const unsigned char *a0 = Vs + row0 * 128 + ((chunk ^ (((row0 >> 1) & 3) << 1)) << 4) + (trp & 1) * 8;
                const int row1 = row0 + 16;
                const unsigned char *a1 = Vs + row1 * 128 + ((chunk ^ (((row1 >> 1) & 3) << 1)) << 4) + (trp & 1) * 8;
                const att_s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((att_lds_s4 *)a0);
                const att_s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((att_lds_s4 *)a1);
                typedef short s8_t __attribute__((ext_vector_type(8)));
                const s8_t vs8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                const f16x8 vf = __builtin_bit_cast(f16x8, vs8);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, o[dt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // O^T accumulators hold 4 consecutive d for one query per lane.  Writing them straight out would be
        // 8-byte stores touching 16 partial lines per instruction (store-issue-bound); instead the wave's
        // 16 x 64 tile goes through its private LDS patch and leaves as whole 128-byte rows, 16 B per lane.
        {
            const float inv = 1.0f / sum;
            _Float16 *ot = Ot + wave * (16 * OS);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                h4_t ov = {(_Float16)(o[dt][0] * inv), (_Float16)(o[dt][1] * inv), (_Float16)(o[dt][2] * inv),
                           (_Float16)(o[dt][3] * inv)};
                *reinterpret_cast<h4_t *>(ot + fr * OS + dt * 16 + fq * 4) = ov;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int row = it * 8 + (lane >> 3), ch = lane & 7;
                const uint4 v = *reinterpret_cast<const uint4 *>(ot + row * OS + ch * 8);
                const int qrow = qt * 16 + row;
                if (qrow < L && !(dbg & 4))
                    store_nt16(out + ((int64_t)b * L + qrow) * W + h * 64 + ch * 8, v);
            }
            __builtin_amdgcn_wave_barrier();
        }
        } // query tiles
    }     // (image, head) pairs
}

// ---------------------------------------------------------------------------------------------
// attention of the `split` precision mode: the same "key on the lane" scheme, fp32-grade on the fp16 matrix cores.
// q | k | v arrive as fp32 [M][3W] (GE_S_BIAS_F32); K and V are split into fp16 pairs (hi, lo) while they are staged into
// LDS (Kh, Kl, Vh, Vl: four arrays in the layouts of the fp16 kernel), Q in registers, and every product runs as
// hi.hi' + lo.hi' + hi.lo' into the one fp32 accumulator:
//   S^T = Kh Qh^T + Kl Qh^T + Kh Ql^T          O^T = Vh^T Ph^T + Vl^T Ph^T + Vh^T Pl^T
// The softmax is fp32 in registers as before; P carries a factor 2^10 (added to the exponent, cancelled by the
// division by the row sum, which carries it too) so that the lo parts of small probabilities stay far away from the
// bottom of the fp16 range.  O leaves as the fp16 pair [hi(W) | lo(W)] per row: the A operand of the out-proj GEMM.
// One workgroup per CU (80-128 KB of LDS); persistent over (image, head) pairs with the next pair's K / V prefetched
// into registers, like the fp16 kernel.
// ---------------------------------------------------------------------------------------------
typedef float att_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 load_nt_f4(const float *p) {
    const att_f4 v = __builtin_nontemporal_load(reinterpret_cast<const att_f4 *>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
}
typedef unsigned att_u4 __attribute__((ext_vector_type(4)));
// 16 bytes through a buffer descriptor: base in scalar registers, 32-bit lane offset, scalar offset (both in bytes); nt
template <typename RS>
__device__ __forceinline__ float4 buffer_load_nt_f4(const RS &rs, unsigned voff, int soff) {
    const att_f4 v = __builtin_bit_cast(att_f4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, soff, 2));
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void split4(const float4 &v, h4_t &hi, h4_t &lo) {
    const float x[4] = {v.x, v.y, v.z, v.w};
    split_pairs<4>(x, hi, lo);   // (common.h: three instructions per two values)
}
#define ATS_FOR_EACH_ITER(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define ATS_DECL(i) float4 k##i = make_float4(0.f, 0.f, 0.f, 0.f), v##i = k##i; unsigned koff##i = 0u;
// thread -> (row, 16-byte chunk of 4 floats), 16 lanes per 256-byte row; rows past L clamped (see the fp16 kernel).
// The byte offset of a thread's chunk inside an (image, head) slab does not depend on the pair: computed ONCE (ATS_OFFS, a
// 32-bit register per iteration) and every load is a BUFFER load: descriptor of the image's q | k | v slab (scalar registers,
// rebuilt per pair from uniform values), 32-bit lane offset, scalar offset of the head's K / V columns -- no vector ALU work per
// load (the 64-bit per-lane address arithmetic of plain global loads cost three v_lshl_add_u64 each).
#define ATS_OFFS(i)                                                                                  \
    if constexpr (K_ITERS > i) {                                                                     \
        const int idx_ = tid + i * NT;                                                               \
        const int row_ = idx_ >> 4, c_ = idx_ & 15;                                                  \
        const int rc_ = row_ < L ? row_ : L - 1;                                                     \
        koff##i = (unsigned)(rc_ * (int)ld + c_ * 4) * 4u;                                           \
    }
#define ATS_LOAD(i)                                                                                  \
    if constexpr (K_ITERS > i) {                                                                     \
        k##i = buffer_load_nt_f4(prs_, koff##i, pso_);                                               \
        v##i = buffer_load_nt_f4(prs_, koff##i, pso_ + W * 4);                                       \
    }
#define ATS_STORE(i)                                                                                 \
    if constexpr (K_ITERS > i) {                                                                     \
        const int idx_ = tid + i * NT;                                                               \
        const int row_ = idx_ >> 4, c_ = idx_ & 15;                                                  \
        if (idx_ < KEYS * 16) {                                                                      \
            h4_t hi_, lo_;                                                                           \
            const int ko_ = row_ * 128 + (((c_ >> 1) ^ (row_ & 7)) << 4) + (c_ & 1) * 8;             \
            split4(k##i, hi_, lo_);                                                                  \
            *reinterpret_cast<h4_t *>(Kh + ko_) = hi_;                                               \
            *reinterpret_cast<h4_t *>(Kl + ko_) = lo_;                                               \
            const int vo_ = row_ * 128 + (((c_ >> 1) ^ (((row_ >> 1) & 3) << 1)) << 4) + (c_ & 1) * 8; \
            split4(v##i, hi_, lo_);                                                                  \
            *reinterpret_cast<h4_t *>(Vh + vo_) = hi_;                                               \
            *reinterpret_cast<h4_t *>(Vl + vo_) = lo_;                                               \
        }                                                                                            \
    }
#define ATS_PREFETCH(pr)                                                                             \
    {                                                                                                \
        const int pb_ = (pr) / heads, ph_ = (pr) - pb_ * heads;                                      \
        const auto prs_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(qkv + (int64_t)pb_ * L * ld), 0,   \
                                                             (int)(L * (int)ld * 4), 0x00020000);   \
        const int pso_ = (ph_ * 64 + W) * 4;                                                         \
        ATS_FOR_EACH_ITER(ATS_LOAD)                                                                  \
    }

// XKEY (L == 16 * KTP + 1, i.e. ViT-B/16 at 256 x 128: L = 129 = 8 key tiles + ONE token): the tiles cover keys 0 .. L-2
// exactly and the last token is handled as an extra key OUTSIDE the matrix instructions -- its K / V rows sit in LDS as
// fp32; its score is 16 fp32 FMAs per lane on the raw Q values + one cross-group sum, its P x V contribution 16 FMAs on the
// O accumulators.  Without it the one odd key costs two of ten key tiles (the PV step is 32 keys wide): 20 % of the QK
// products, a fifth PV step, 20 % of the softmax and 16 KB of LDS -- and with 64 + 1 KB of K / V pairs instead of 80 a
// 256-thread workgroup fits a CU TWICE, so one workgroup's loads / staging / barriers overlap the other's compute.
template <int KTP, int NW, bool XKEY>
__global__ __launch_bounds__(64 * NW, (NW >= 8 || XKEY) ? 2 : 1) void attention_split_kernel(const float *__restrict__ qkv, int L, int W,
                                                                                   int heads, _Float16 *__restrict__ out,
                                                                                   int q_tiles, int total_pairs, int dbg_) {
    // The timing-ablation switches are compile-time zero in the product build: as a run-time argument `dbg & 32` kept
    // every key tile's score chain behind its own branch (eight basic blocks, each one strictly serial chain of six
    // dependent matrix instructions with the full issue-to-result latency between them) -- with the flag gone the eight
    // independent chains are one block and interleave.
#ifdef MPREID_ABLATION
    const int dbg = dbg_;
#else
    constexpr int dbg = 0;
    (void)dbg_;
#endif
    constexpr int KEYS = KTP * 16;
    constexpr int NT = 64 * NW;
    constexpr int K_ITERS = (KEYS * 16 + NT - 1) / NT;
    static_assert(K_ITERS <= 8, "extend ATS_FOR_EACH_ITER");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *Kh = smem, *Kl = smem + KEYS * 128, *Vh = smem + 2 * KEYS * 128, *Vl = smem + 3 * KEYS * 128;
    constexpr int OS = 72;
    _Float16 *Ot = reinterpret_cast<_Float16 *>(smem + 4 * KEYS * 128);   // [NW][16][OS]
    float *Xk = reinterpret_cast<float *>(smem + 4 * KEYS * 128 + NW * 16 * OS * 2);   // XKEY: [64] K row, [64] V row of token L-1 (fp32)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int64_t ld = 3 * (int64_t)W;
    const int nqt = (q_tiles > 0) ? q_tiles : (L + 15) / 16;
    const float scale_log2e = 0.125f * 1.44269504088896340736f;

    ATS_FOR_EACH_ITER(ATS_DECL)
    ATS_FOR_EACH_ITER(ATS_OFFS)

    float4 xrow = make_float4(0.f, 0.f, 0.f, 0.f);   // XKEY: lanes 0-15 of wave 0 hold the K row of token L-1, lanes 16-31 its V row
    auto x_request = [&](int pr) {
        if constexpr (XKEY) {
            if (tid < 32) {
                const int pb = pr / heads, ph = pr - pb * heads;
                xrow = load_nt_f4(qkv + ((int64_t)pb * L + (L - 1)) * ld + ph * 64 + (tid < 16 ? W : 2 * W) + (tid & 15) * 4);
            }
        }
    };
    // dbg (MPREID_ATT_DBG in an MPREID_ABLATION build; timing ablations only, wrong results): 1 no K / V loads, 2 no compute,
    // 4 no output stores, 8 no PV, 16 no O write-out, 32 no QK products.  Round-3 ablation at B = 508, L = 129 (us per
    // layer-batch): full 263; loads + staging only 100; compute on stale LDS 220 = QK 40 + PV 60 + O write-out 26 + the
    // rest 102 (softmax and the P pairs ~60: ~400 VALU instructions per tile, K / V staging, two barriers per pair, the
    // 9 tiles over 8 waves).
    int pair = blockIdx.x;
    if (pair >= total_pairs) return;
    if (!(dbg & 1)) {
        ATS_PREFETCH(pair)
        x_request(pair);
    }
    for (; pair < total_pairs; pair += gridDim.x) {
        const int b = pair / heads, h = pair - b * heads;
        const float *base = qkv + (int64_t)b * L * ld + h * 64;
        __syncthreads(); // every wave is done reading the previous pair from LDS
        ATS_FOR_EACH_ITER(ATS_STORE)
        if constexpr (XKEY) {
            if (tid < 32) *reinterpret_cast<float4 *>(Xk + tid * 4) = xrow;
        }
        // Q of this wave's first query tile (B operand: B[k = d][col = query]; 8 consecutive d per lane and 32-wide k step),
        // requested BEFORE the barrier so that its round trip is hidden behind it; the next tile's Q (a wave has a second
        // tile only when there are more tiles than waves) is requested while the current tile is computed.  Fetched at the
        // top of each tile instead, every tile started with an exposed global round trip: 289 -> 2xx us per layer-batch.
        float4 qraw[4];
        auto q_request = [&](int qt) {
            const int q = qt * 16 + fr;
            const int qc = q < L ? q : L - 1;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const float *qp = base + (int64_t)qc * ld + ks * 32 + fq * 8;
                qraw[2 * ks] = *reinterpret_cast<const float4 *>(qp);
                qraw[2 * ks + 1] = *reinterpret_cast<const float4 *>(qp + 4);
            }
        };
        q_request(wave < nqt ? wave : 0);
        __syncthreads();
        {
            const int nxt = pair + (int)gridDim.x < total_pairs ? pair + (int)gridDim.x : pair;
            if (!(dbg & 1)) {
                ATS_PREFETCH(nxt)
                x_request(nxt);
            }
        }
        // XKEY, all query tiles wanted: the tiles 0 .. KTP-1 are full, tile KTP holds ONE query (token L-1).  Handed to one
        // wave as a tile of its own (round 3) that wave did three tiles per pair where the others did two, and everybody
        // waited for it at the next barrier: a third of the compute phase for 1/129 of the rows.  Now the waves share it BY
        // KEYS (below, after the main tiles): every wave scores the query against its own two key tiles, the partial
        // (max, sum, O) leave through LDS and wave 0 merges them -- a quarter tile per wave instead of a whole one for one.
        const bool tail = XKEY && nqt == KTP + 1;
        const int nmain = tail ? KTP : nqt;
        for (int qt = wave; qt < ((dbg & 2) ? 0 : nmain); qt += NW) {
            f16x8 qh[2], ql[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const float4 a0 = qraw[2 * ks], a1 = qraw[2 * ks + 1];
                const float qv[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                split_pairs<8>(qv, qh[ks], ql[ks]);
            }
            float sx = 0.f;   // XKEY: q . k of the extra key (token L-1), fp32, for this lane's query column
            if constexpr (XKEY) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {   // this lane's 16 d values: ks * 32 + fq * 8 + (c & 1) * 4 .. + 3
                    const float4 kx = *reinterpret_cast<const float4 *>(Xk + (c >> 1) * 32 + fq * 8 + (c & 1) * 4);
                    sx = fmaf(qraw[c].x, kx.x, sx);
                    sx = fmaf(qraw[c].y, kx.y, sx);
                    sx = fmaf(qraw[c].z, kx.z, sx);
                    sx = fmaf(qraw[c].w, kx.w, sx);
                }
                sx = xor16_32_sum(sx);
            }
            if (qt + NW < nmain) q_request(qt + NW);
            else if (tail) q_request(KTP);   // (every column of that tile clamps to token L-1: sixteen copies of the one query)
            f32x4 s[KTP];
            float mx = -3.0e38f;
            const int nkt = XKEY ? KTP : (L + 15) >> 4;   // key tiles that hold a valid key
#pragma unroll
            for (int kt = 0; kt < KTP; ++kt) {
                s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (kt < nkt && !(dbg & 32)) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const int ko = (kt * 16 + fr) * 128 + (((ks * 4 + fq) ^ (lane & 7)) << 4);
                        const f16x8 kh = *reinterpret_cast<const f16x8 *>(Kh + ko);
                        const f16x8 kl = *reinterpret_cast<const f16x8 *>(Kl + ko);
                        s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, qh[ks], s[kt], 0, 0, 0);
                        s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl, qh[ks], s[kt], 0, 0, 0);
                        s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, ql[ks], s[kt], 0, 0, 0);
                    }
                }
                if (kt & 1) __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int kt = 0; kt < KTP; ++kt) {
                if constexpr (!XKEY) {   // (XKEY: every key of the tiles is valid)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kt * 16 + fq * 4 + r >= L) s[kt][r] = -3.0e38f;
                }
                mx = fmaxf(mx, fmaxf(fmaxf(s[kt][0], s[kt][1]), fmaxf(s[kt][2], s[kt][3])));
            }
            mx = xor16_32_max(mx);
            if constexpr (XKEY) mx = fmaxf(mx, sx);
            // p * 2^10 = exp2(s*c - mx*c + 10); masked entries give exp2(-huge) = 0
            const float nmx = fmaf(-mx, scale_log2e, 10.0f);
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < KTP; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(s[kt][r], scale_log2e, nmx));
                    s[kt][r] = p;
                    sum += p;
                }
            sum = xor16_32_sum(sum);
            float px = 0.f;
            if constexpr (XKEY) {
                px = __builtin_amdgcn_exp2f(fmaf(sx, scale_log2e, nmx));
                sum += px;
            }
            f32x4 o[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s2 = 0; s2 < ((dbg & 8) ? 0 : KTP / 2); ++s2) {
                f16x8 ph, pl;
                {
                    const float pv[8] = {s[2 * s2][0], s[2 * s2][1], s[2 * s2][2], s[2 * s2][3],
                                         s[2 * s2 + 1][0], s[2 * s2 + 1][1], s[2 * s2 + 1][2], s[2 * s2 + 1][3]};
                    split_pairs<8>(pv, ph, pl);
                }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const int trq = fr >> 2, trp = fr & 3;
                    const int row0 = s2 * 32 + fq * 4 + trq, row1 = row0 + 16;
                    const int chunk = dt * 2 + (trp >> 1);
                    const int o0 = row0 * 128 + ((chunk ^ (((row0 >> 1) & 3) << 1)) << 4) + (trp & 1) * 8;
                    const int o1 = row1 * 128 + ((chunk ^ (((row1 >> 1) & 3) << 1)) << 4) + (trp & 1) * 8;
                    typedef short s8_t __attribute__((ext_vector_type(8)));
                    const att_s4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((att_lds_s4 *)(Vh + o0));
                    const att_s4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((att_lds_s4 *)(Vh + o1));
                    const att_s4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((att_lds_s4 *)(Vl + o0));
                    const att_s4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((att_lds_s4 *)(Vl + o1));
                    const s8_t vh8 = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                    const s8_t vl8 = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
                    const f16x8 vh = __builtin_bit_cast(f16x8, vh8), vl = __builtin_bit_cast(f16x8, vl8);
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, ph, o[dt], 0, 0, 0);
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, ph, o[dt], 0, 0, 0);
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, pl, o[dt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (XKEY) {   // + p_x * v_x: lane (query fr) owns d = dt * 16 + fq * 4 .. + 3
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const float4 vx = *reinterpret_cast<const float4 *>(Xk + 64 + dt * 16 + fq * 4);
                    o[dt][0] = fmaf(px, vx.x, o[dt][0]);
                    o[dt][1] = fmaf(px, vx.y, o[dt][1]);
                    o[dt][2] = fmaf(px, vx.z, o[dt][2]);
                    o[dt][3] = fmaf(px, vx.w, o[dt][3]);
                }
            }
            if (!(dbg & 16)) {   // O^T -> fp16 pair, through the wave's LDS patch as whole 128-byte rows: hi part, then lo part
                const float inv = 1.0f / sum;
                _Float16 *ot = Ot + wave * (16 * OS);
                h4_t ohi[4], olo[4];   // both halves once (split_pairs: three instructions per two values)
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const float ovf[4] = {o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv};
                    split_pairs<4>(ovf, ohi[dt], olo[dt]);
                }
#pragma unroll
                for (int part = 0; part < 2; ++part) {
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt)
                        *reinterpret_cast<h4_t *>(ot + fr * OS + dt * 16 + fq * 4) = part == 0 ? ohi[dt] : olo[dt];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        const int row = it * 8 + (lane >> 3), ch = lane & 7;
                        const uint4 v = *reinterpret_cast<const uint4 *>(ot + row * OS + ch * 8);
                        const int qrow = qt * 16 + row;
                        if (qrow < L && !(dbg & 4)) store_nt16(out + ((int64_t)b * L + qrow) * 2 * W + part * W + h * 64 + ch * 8, v);
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        } // query tiles
        if constexpr (XKEY) {
            if (tail && !(dbg & 2)) {
                static_assert(!XKEY || KTP == 2 * NW, "the tail query is shared by keys: two key tiles (one PV step) per wave");
                f16x8 qh[2], ql[2];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const float4 a0 = qraw[2 * ks], a1 = qraw[2 * ks + 1];
                    const float qv[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                    split_pairs<8>(qv, qh[ks], ql[ks]);
                }
                f32x4 s2t[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int kt = 2 * wave + t;
                    s2t[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const int ko = (kt * 16 + fr) * 128 + (((ks * 4 + fq) ^ (lane & 7)) << 4);
                        const f16x8 kh = *reinterpret_cast<const f16x8 *>(Kh + ko);
                        const f16x8 kl = *reinterpret_cast<const f16x8 *>(Kl + ko);
                        s2t[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, qh[ks], s2t[t], 0, 0, 0);
                        s2t[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl, qh[ks], s2t[t], 0, 0, 0);
                        s2t[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, ql[ks], s2t[t], 0, 0, 0);
                    }
                }
                float mx = fmaxf(fmaxf(fmaxf(s2t[0][0], s2t[0][1]), fmaxf(s2t[0][2], s2t[0][3])),
                                 fmaxf(fmaxf(s2t[1][0], s2t[1][1]), fmaxf(s2t[1][2], s2t[1][3])));
                mx = xor16_32_max(mx);
                float sx = 0.f;
                if (wave == 0) {   // the extra key (token L-1 as a KEY) belongs to wave 0's share
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float4 kx = *reinterpret_cast<const float4 *>(Xk + (c >> 1) * 32 + fq * 8 + (c & 1) * 4);
                        sx = fmaf(qraw[c].x, kx.x, sx);
                        sx = fmaf(qraw[c].y, kx.y, sx);
                        sx = fmaf(qraw[c].z, kx.z, sx);
                        sx = fmaf(qraw[c].w, kx.w, sx);
                    }
                    sx = xor16_32_sum(sx);
                    mx = fmaxf(mx, sx);
                }
                const float nmx = fmaf(-mx, scale_log2e, 10.0f);
                float sum = 0.f;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = __builtin_amdgcn_exp2f(fmaf(s2t[t][r], scale_log2e, nmx));
                        s2t[t][r] = pv;
                        sum += pv;
                    }
                sum = xor16_32_sum(sum);
                float px = 0.f;
                if (wave == 0) {
                    px = __builtin_amdgcn_exp2f(fmaf(sx, scale_log2e, nmx));
                    sum += px;
                }
                f32x4 o[4];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
                {
                    f16x8 ph, pl;
                    {
                        const float pv[8] = {s2t[0][0], s2t[0][1], s2t[0][2], s2t[0][3], s2t[1][0], s2t[1][1], s2t[1][2], s2t[1][3]};
                        split_pairs<8>(pv, ph, pl);
                    }
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        const int trq = fr >> 2, trp = fr & 3;
                        const int row0 = wave * 32 + fq * 4 + trq, row1 = row0 + 16;
                        const int chunk = dt * 2 + (trp >> 1);
                        const int o0 = row0 * 128 + ((chunk ^ (((row0 >> 1) & 3) << 1)) << 4) + (trp & 1) * 8;
                        const int o1 = row1 * 128 + ((chunk ^ (((row1 >> 1) & 3) << 1)) << 4) + (trp & 1) * 8;
                        typedef short s8_t __attribute__((ext_vector_type(8)));
                        const att_s4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((att_lds_s4 *)(Vh + o0));
                        const att_s4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((att_lds_s4 *)(Vh + o1));
                        const att_s4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((att_lds_s4 *)(Vl + o0));
                        const att_s4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((att_lds_s4 *)(Vl + o1));
                        const s8_t vh8 = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                        const s8_t vl8 = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
                        const f16x8 vh = __builtin_bit_cast(f16x8, vh8), vl = __builtin_bit_cast(f16x8, vl8);
                        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, ph, o[dt], 0, 0, 0);
                        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, ph, o[dt], 0, 0, 0);
                        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, pl, o[dt], 0, 0, 0);
                    }
                }
                if (wave == 0) {
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        const float4 vx = *reinterpret_cast<const float4 *>(Xk + 64 + dt * 16 + fq * 4);
                        o[dt][0] = fmaf(px, vx.x, o[dt][0]);
                        o[dt][1] = fmaf(px, vx.y, o[dt][1]);
                        o[dt][2] = fmaf(px, vx.z, o[dt][2]);
                        o[dt][3] = fmaf(px, vx.w, o[dt][3]);
                    }
                }
                // partials of query column 0 (= token L-1; the other fifteen columns are copies): the wave's own patch holds
                // O[64] | max | sum as fp32
                float *pt = reinterpret_cast<float *>(Ot + wave * (16 * OS));
                if (fr == 0) {
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt)
                        *reinterpret_cast<float4 *>(pt + dt * 16 + fq * 4) = make_float4(o[dt][0], o[dt][1], o[dt][2], o[dt][3]);
                    if (fq == 0) {
                        pt[64] = mx;
                        pt[65] = sum;
                    }
                }
                __syncthreads();
                if (wave == 0) {   // merge in wave order (fixed: the result does not depend on timing), lane = output dimension d
                    float mw[NW], lw[NW], M = -3.0e38f;
#pragma unroll
                    for (int w2 = 0; w2 < NW; ++w2) {
                        const float *pw = reinterpret_cast<const float *>(Ot + w2 * (16 * OS));
                        mw[w2] = pw[64];
                        lw[w2] = pw[65];
                        M = fmaxf(M, mw[w2]);
                    }
                    float den = 0.f, num = 0.f;
#pragma unroll
                    for (int w2 = 0; w2 < NW; ++w2) {
                        const float *pw = reinterpret_cast<const float *>(Ot + w2 * (16 * OS));
                        const float sc = __builtin_amdgcn_exp2f((mw[w2] - M) * scale_log2e);
                        den = fmaf(lw[w2], sc, den);
                        num = fmaf(pw[lane], sc, num);
                    }
                    const float v = num / den;
                    const _Float16 hi = (_Float16)v;
                    const _Float16 lo = (_Float16)(v - (float)hi);
                    if (!(dbg & 4)) {
                        _Float16 *orow = out + ((int64_t)b * L + (L - 1)) * 2 * W + h * 64 + lane;
                        orow[0] = hi;
                        orow[W] = lo;
                    }
                }
            }
        }
    }     // (image, head) pairs
}

// ---------------------------------------------------------------------------------------------
// head: ln_post on the CLS row, CLS @ proj, optional eval-BN necks, concat -> [B][W + out_dim]
// model/clip/model.py:471-474, model/make_model.py:98-115
// ---------------------------------------------------------------------------------------------
constexpr int HEAD_IMGS = 2;   // (8 images per workgroup: 128 workgroups, 90 us at B = 508; 2: 508 workgroups)
// ln_post of the CLS rows (one wave per row): y[b][:] (fp32, for the projection) and out[b][0:W]
__global__ __launch_bounds__(256) void cls_ln_kernel(const float *__restrict__ x, int64_t row_stride, int B, int W,
                                                     int out_dim, const float *__restrict__ g,
                                                     const float *__restrict__ bta, const float *__restrict__ bn_s,
                                                     const float *__restrict__ bn_b, float *__restrict__ y,
                                                     float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float *xr = x + (int64_t)b * row_stride;
    float s = 0.f;
    for (int k = lane; k < W; k += 64) s += xr[k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    const float mean = s / (float)W;
    float q = 0.f;
    for (int k = lane; k < W; k += 64) {
        const float d = xr[k] - mean;
        q += d * d;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) q += __shfl_xor(q, off, 64);
    const float rstd = 1.0f / sqrtf(q / (float)W + 1e-5f);
    float *orow = out + (int64_t)b * (W + out_dim);
    for (int k = lane; k < W; k += 64) {
        const float v = (xr[k] - mean) * rstd * g[k] + bta[k];
        y[(int64_t)b * W + k] = v;
        orow[k] = bn_s ? fmaf(v, bn_s[k], bn_b[k]) : v;
    }
}

// out[b][W + o] = sum_k y[b][k] * proj[k][o]  (fp32, k ascending).  grid = (ceil(B/8), ceil(out_dim/256));
// 8 rows of y sit in LDS, every thread owns one output column for those 8 images.
__global__ __launch_bounds__(256) void cls_proj_kernel(const float *__restrict__ y, int B, int W, int out_dim,
                                                       const float *__restrict__ proj,
                                                       const float *__restrict__ bnp_s,
                                                       const float *__restrict__ bnp_b, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *ys = reinterpret_cast<float *>(smem); // [HEAD_IMGS][W]
    const int tid = threadIdx.x;
    const int b0 = blockIdx.x * HEAD_IMGS;
    for (int idx = tid; idx < HEAD_IMGS * W; idx += 256) {
        const int i = idx / W, k = idx - i * W;
        ys[idx] = (b0 + i < B) ? y[(int64_t)(b0 + i) * W + k] : 0.f;
    }
    __syncthreads();
    const int o = blockIdx.y * 256 + tid;
    if (o >= out_dim) return;
    float acc[HEAD_IMGS];
#pragma unroll
    for (int i = 0; i < HEAD_IMGS; ++i) acc[i] = 0.f;
#pragma unroll 8
    for (int k = 0; k < W; ++k) {
        const float pw = proj[(int64_t)k * out_dim + o];
#pragma unroll
        for (int i = 0; i < HEAD_IMGS; ++i) acc[i] = fmaf(ys[i * W + k], pw, acc[i]);
    }
#pragma unroll
    for (int i = 0; i < HEAD_IMGS; ++i) {
        const int b = b0 + i;
        if (b < B) out[(int64_t)b * (W + out_dim) + W + o] = bnp_s ? fmaf(acc[i], bnp_s[o], bnp_b[o]) : acc[i];
    }
}

// CLS rows of the residual stream / attention output -> compact [Bpad][W] buffers (last block only)
// (aw = row length of a in halfs: W, or 2W for the fp16 pairs of the split mode)
__global__ __launch_bounds__(256) void gather_cls_kernel(const float *__restrict__ x, const _Float16 *__restrict__ a,
                                                         int B, int L, int W, int aw, float *__restrict__ x_cls,
                                                         _Float16 *__restrict__ a_cls) {
    const int b = blockIdx.x;
    for (int k = threadIdx.x; k < W; k += 256) x_cls[(int64_t)b * W + k] = x[(int64_t)b * L * W + k];
    for (int k = threadIdx.x; k < aw; k += 256) a_cls[(int64_t)b * aw + k] = a[(int64_t)b * L * aw + k];
}

// ---------------------------------------------------------------------------------------------
// host driver
// ---------------------------------------------------------------------------------------------
struct VitLayout {
    int L, P, M, Mpad, MPpad, Kp, Bpad;
    size_t patches, x, a, qkv, hbuf, x_cls, a_cls, h_cls, y_cls, total;
};

static VitLayout vit_layout(const mpreid_vit_cfg *c, int B) {
    VitLayout v{};
    v.P = c->h_res * c->w_res;
    v.L = v.P + 1;
    v.M = B * v.L;
    v.Mpad = (int)align_up((size_t)v.M, 256);   // 256-row tiles of the big GEMM kernel
    v.MPpad = (int)align_up((size_t)B * v.P, 256);
    v.Kp = 3 * c->patch * c->patch;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += align_up(bytes, 256);
        return o;
    };
    const size_t W = (size_t)c->width;
    // split precision mode: every GEMM operand is an fp16 pair (2x the halfs per row), q | k | v are fp32
    const size_t pr = c->precision != MPREID_VIT_F16 ? 2 : 1;
    v.patches = take((size_t)v.MPpad * v.Kp * 2 * pr);
    v.x = take((size_t)v.Mpad * W * 4);
    v.a = take((size_t)v.Mpad * W * 2 * pr);
    v.qkv = take((size_t)v.Mpad * 3 * W * 2 * pr);
    v.hbuf = take((size_t)v.Mpad * 4 * W * 2 * pr);
    v.Bpad = (int)align_up((size_t)B, GBM);
    v.x_cls = take((size_t)v.Bpad * W * 4);
    v.a_cls = take((size_t)v.Bpad * W * 2 * pr);
    v.h_cls = take((size_t)v.Bpad * 4 * W * 2 * pr);
    v.y_cls = take((size_t)v.Bpad * W * 4);
    v.total = off;
    return v;
}

static int vit_check_cfg(const mpreid_vit_cfg *c) {
    ARG_CHECK(c != nullptr);
    ARG_CHECK(c->width > 0 && c->heads > 0 && c->layers >= 0 && c->out_dim > 0);
    if (c->precision == 2) {   // MPREID_VIT_SPLIT_LNFOLD of rounds 3: see include/mpreid.h
        mpreid_set_error("the folded-LayerNorm form of the split mode was removed in round 4 (not faster than MPREID_VIT_SPLIT, and not "
                         "reproducible run to run at small batches); use MPREID_VIT_SPLIT");
        return MPREID_ERR_UNSUPPORTED;
    }
    ARG_CHECK(c->precision == MPREID_VIT_F16 || c->precision == MPREID_VIT_SPLIT);
    if (c->width != c->heads * 64) {
        mpreid_set_error("head dim must be 64 (width %d, heads %d)", c->width, c->heads);
        return MPREID_ERR_UNSUPPORTED;
    }
    if (c->width % 128 != 0 || c->width > 1024 || (3 * c->patch * c->patch) % 64 != 0 || c->patch % 8 != 0) {
        mpreid_set_error("unsupported ViT geometry (width %d, patch %d)", c->width, c->patch);
        return MPREID_ERR_UNSUPPORTED;
    }
    ARG_CHECK(c->h_res == (c->img_h - c->patch) / c->stride + 1 && c->w_res == (c->img_w - c->patch) / c->stride + 1);
    if (c->h_res * c->w_res + 1 > 256) {
        mpreid_set_error("token count %d > 256 not supported by the LDS-resident attention kernel",
                         c->h_res * c->w_res + 1);
        return MPREID_ERR_UNSUPPORTED;
    }
    return MPREID_OK;
}

extern "C" size_t mpreid_vit_workspace_bytes(const mpreid_vit_cfg *cfg, int batch) {
    if (!cfg || batch <= 0) return 0;
    return vit_layout(cfg, batch).total;
}

template <int KTP, int NW, bool EXACT>
static int launch_attention(const _Float16 *qkv, int B, int L, int W, int heads, _Float16 *out, int q_tiles,
                            hipStream_t stream) {
    constexpr int KEYS = KTP * 16;
    const size_t lds = (size_t)2 * KEYS * 128 + (size_t)NW * 16 * 72 * 2;
    // per-device state (a process may drive several GPUs: the reference's do_inference is multi-device in one
    // process, processor/processor.py:178-182): the dynamic-LDS attribute and the occupancy are set / queried once
    // per HIP device, under a mutex
    constexpr int MAX_DEV = 64;
    static std::mutex mu;
    static int blocks_per_cu_dev[MAX_DEV] = {0};
    int dev = 0, cus = 256;
    HIP_TRY(hipGetDevice(&dev));
    int blocks_per_cu;
    {
        std::lock_guard<std::mutex> lk(mu);
        const int slot = dev < MAX_DEV ? dev : MAX_DEV - 1;
        if (blocks_per_cu_dev[slot] == 0 || dev >= MAX_DEV) {
            if (lds > 48 * 1024)
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(attention_kernel<KTP, NW, EXACT>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            // persistent grid: as many workgroups as fit the chip at once (LDS and the 4-waves/SIMD register
            // bound), each walking pairs with stride gridDim
            int occ = 0;
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(
                &occ, reinterpret_cast<const void *>(attention_kernel<KTP, NW, EXACT>), 64 * NW, lds));
            blocks_per_cu_dev[slot] = occ > 0 ? occ : 1;
            if (mpreid_tune("verbose", 0))
                fprintf(stderr, "[mpreid] attention<%d,%d> on device %d: %d workgroups/CU by the occupancy API (lds %zu B, %d threads)\n",
                        KTP, NW, dev, occ, lds, 64 * NW);
        }
        blocks_per_cu = blocks_per_cu_dev[slot];
    }
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int total = B * heads;
    int grid = cus * blocks_per_cu;
    if (grid > total) grid = total;
    hipLaunchKernelGGL((attention_kernel<KTP, NW, EXACT>), dim3((unsigned)grid), dim3(64 * NW), lds, stream, qkv, L, W,
                       heads, out, q_tiles, total, mpreid_ablation_env("MPREID_ATT_DBG"));
    LAUNCH_CHECK();
    return MPREID_OK;
}

template <int KTP, int NW, bool XKEY = false>
static int launch_attention_split(const float *qkv, int B, int L, int W, int heads, _Float16 *out, int q_tiles,
                                  hipStream_t stream) {
    constexpr int KEYS = KTP * 16;
    const size_t lds = (size_t)4 * KEYS * 128 + (size_t)NW * 16 * 72 * 2 + (XKEY ? 512 : 0);
    static PerDeviceOnce attr_once;
    const int rc = attr_once.run([&]() -> int {
        if (lds > 48 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(attention_split_kernel<KTP, NW, XKEY>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        return MPREID_OK;
    });
    if (rc) return rc;
    int dev = 0, cus = 256;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int total = B * heads;
    // persistent: as many workgroups as the LDS lets a CU hold (one for L = 129; the small test shapes fit several)
    int per_cu = (int)((160 * 1024) / lds);
    per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
    int grid = cus * per_cu;
    if (grid > total) grid = total;
    static const int att_dbg = mpreid_ablation_env("MPREID_ATT_DBG");
    hipLaunchKernelGGL((attention_split_kernel<KTP, NW, XKEY>), dim3((unsigned)grid), dim3(64 * NW), lds, stream, qkv, L, W, heads,
                       out, q_tiles, total, att_dbg);
    LAUNCH_CHECK();
    return MPREID_OK;
}

static int attention_split_dispatch(const float *qkv, int B, int L, int W, int heads, _Float16 *out, int q_tiles,
                                    hipStream_t stream) {
    const int kt = (L + 15) / 16;
    if (kt <= 2) return launch_attention_split<2, 2>(qkv, B, L, W, heads, out, q_tiles, stream);
    if (L == 129) return launch_attention_split<8, 4, true>(qkv, B, L, W, heads, out, q_tiles, stream);   // ViT-B/16 at 256 x 128
    if (kt <= 10) return launch_attention_split<10, 8>(qkv, B, L, W, heads, out, q_tiles, stream);
    if (kt <= 14) return launch_attention_split<14, 8>(qkv, B, L, W, heads, out, q_tiles, stream);
    return launch_attention_split<16, 8>(qkv, B, L, W, heads, out, q_tiles, stream);
}

// waves per workgroup: one per 16-query tile when the block can hold them; at least 4 so that the
// K/V staging of a CLS-only call (one query tile) is still spread over 256 threads
static int attention_dispatch(const _Float16 *qkv, int B, int L, int W, int heads, _Float16 *out, int q_tiles,
                              hipStream_t stream) {
    const int kt = (L + 15) / 16;
    if (kt <= 2) return launch_attention<2, 2, true>(qkv, B, L, W, heads, out, q_tiles, stream);
    if (kt <= 10) {
        const bool exact = kt >= 9;
        // (L = 129 = nine query tiles: nine waves, one tile each, measured 123 us against 111 for eight waves with wave 0
        // taking tile 8 as well.  Ablations of the 8-wave kernel, same run: loads + LDS staging alone 61 us, compute on
        // stale LDS alone 80 us, stores 5 us)
        // four waves per workgroup (2-3 query tiles each): 50 KB of LDS = three workgroups per CU and no register
        // spills; eight waves (two workgroups per CU under a 128-VGPR bound, 44 B of scratch per lane, wave 0 taking
        // tile 8 as well) measured 113 us against 105 at B = 508, L = 129
        return exact ? launch_attention<10, 4, true>(qkv, B, L, W, heads, out, q_tiles, stream)
                     : launch_attention<10, 4, false>(qkv, B, L, W, heads, out, q_tiles, stream);
    }
    if (kt <= 14)
        return kt >= 13 ? launch_attention<14, 8, true>(qkv, B, L, W, heads, out, q_tiles, stream)
                        : launch_attention<14, 8, false>(qkv, B, L, W, heads, out, q_tiles, stream);
    return kt >= 15 ? launch_attention<16, 8, true>(qkv, B, L, W, heads, out, q_tiles, stream)
                    : launch_attention<16, 8, false>(qkv, B, L, W, heads, out, q_tiles, stream);
}

static int vit_forward_impl(const mpreid_vit_cfg *cfg, const mpreid_vit_weights *w, const float *img,
                            const unsigned char *img_u8, const float *mean3, const float *std3, int view, int B,
                            const float *cv_emb, float *out, void *ws, size_t ws_bytes, mpreid_stream_t stream_) {
    int rc = vit_check_cfg(cfg);
    if (rc) return rc;
    ARG_CHECK(w && (img || img_u8) && out && B > 0 && w->layers && view >= 0 && view <= 3);
    const VitLayout v = vit_layout(cfg, B);
    if (!ws || ws_bytes < v.total) {
        mpreid_set_error("vit workspace too small: %zu < %zu", ws_bytes, v.total);
        return MPREID_ERR_WORKSPACE;
    }
    hipStream_t stream = (hipStream_t)stream_;
    const int W = cfg->width, L = v.L;
    char *base = (char *)ws;
    _Float16 *patches = (_Float16 *)(base + v.patches);
    float *x = (float *)(base + v.x);
    _Float16 *a = (_Float16 *)(base + v.a);
    _Float16 *qkv = (_Float16 *)(base + v.qkv);
    _Float16 *hbuf = (_Float16 *)(base + v.hbuf);
    float *x_cls = (float *)(base + v.x_cls);
    _Float16 *a_cls = (_Float16 *)(base + v.a_cls);
    _Float16 *h_cls = (_Float16 *)(base + v.h_cls);
    float *y_cls = (float *)(base + v.y_cls);
    const bool cls_last = cfg->cls_only_last != 0 && cfg->layers > 0;
    // split precision mode: every linear layer runs hi.hi' + lo.hi' + hi.lo' over fp16 pairs (gemm_f16.h GE_S_*)
    const bool split = cfg->precision != MPREID_VIT_F16;
    const int pr = split ? 2 : 1;   // halfs per logical element of a GEMM operand row
    auto linear = [&](GemmArgs &g, int kdim, float wscale, int epi_f16, int epi_split) -> int {
        g.K = pr * kdim;
        if (split) {
            g.kseg = kdim;
            g.oscale = wscale;
        }
        return launch_gemm_f16(g, split ? epi_split : epi_f16, stream);
    };

    // patch embedding (conv1, no bias) + positional embedding; CLS row; ln_pre
    {
        const int64_t threads = (int64_t)v.MPpad * (v.Kp / 8);
        const dim3 grid((unsigned)((threads + 255) / 256));
#define MPREID_IM2COL_S(V, S)                                                                                          \
        if (img_u8)                                                                                                    \
            hipLaunchKernelGGL((im2col_u8_kernel<V, S>), grid, dim3(256), 0, stream, img_u8, B, cfg->img_h, cfg->img_w, \
                               cfg->patch, cfg->stride, cfg->h_res, cfg->w_res, mean3[0], mean3[1], mean3[2], std3[0], \
                               std3[1], std3[2], patches, v.MPpad);                                                    \
        else                                                                                                           \
            hipLaunchKernelGGL((im2col_kernel<V, S>), grid, dim3(256), 0, stream, img, B, cfg->img_h, cfg->img_w,      \
                               cfg->patch, cfg->stride, cfg->h_res, cfg->w_res, patches, v.MPpad);
#define MPREID_IM2COL(V)                                                                                               \
    case V:                                                                                                            \
        if (split) { MPREID_IM2COL_S(V, true) } else { MPREID_IM2COL_S(V, false) }                                     \
        break;
        switch (view) {
            MPREID_IM2COL(0) MPREID_IM2COL(1) MPREID_IM2COL(2) MPREID_IM2COL(3)
        }
#undef MPREID_IM2COL
#undef MPREID_IM2COL_S
        LAUNCH_CHECK();
        GemmArgs g{};
        g.A = patches;
        g.W = (const _Float16 *)w->conv_w;
        g.M = v.MPpad;
        g.N = W;
        g.out = x;
        g.ldo = W;
        g.aux = w->pos_emb;
        g.m_valid = B * v.P;
        g.P = v.P;
        g.L = L;
        rc = linear(g, v.Kp, w->conv_s, GE_PATCH, GE_S_PATCH);
        if (rc) return rc;
        hipLaunchKernelGGL(cls_token_kernel, dim3((unsigned)B), dim3(256), 0, stream, w->class_emb, w->pos_emb, cv_emb,
                           B, L, W, x);
        hipLaunchKernelGGL(layernorm_kernel<0>, dim3((unsigned)((v.M + 3) / 4)), dim3(256), 0, stream, x,
                           (int64_t)v.M, W, w->ln_pre_g, w->ln_pre_b, (void *)x, (int64_t)W);
        LAUNCH_CHECK();
    }
    for (int l = 0; l < cfg->layers; ++l) {
        const mpreid_vit_layer &ly = w->layers[l];
        const bool tail = cls_last && (l == cfg->layers - 1);
        GemmArgs g{};
        // x = x + out_proj(attn(ln_1(x)))
        void *ptok = mpreid_prof_begin(stream);
        if (split)
            hipLaunchKernelGGL(layernorm_kernel<2>, dim3((unsigned)((v.M + 3) / 4)), dim3(256), 0, stream, x,
                               (int64_t)v.M, W, ly.ln1_g, ly.ln1_b, (void *)a, (int64_t)W);
        else
            hipLaunchKernelGGL(layernorm_kernel<1>, dim3((unsigned)((v.M + 3) / 4)), dim3(256), 0, stream, x,
                               (int64_t)v.M, W, ly.ln1_g, ly.ln1_b, (void *)a, (int64_t)W);
        mpreid_prof_end(ptok, stream, MPREID_PROF_LAYERNORM, v.M, W, pr, (double)v.M * W * (4.0 + 2.0 * pr));
        LAUNCH_CHECK();
        g = GemmArgs{};
        g.A = a; g.W = (const _Float16 *)ly.in_proj_w; g.M = v.Mpad; g.N = 3 * W;
        g.out = qkv; g.ldo = 3 * W; g.bias = ly.in_proj_b;
        if ((rc = linear(g, W, ly.in_proj_s, GE_BIAS_F16, GE_S_BIAS_F32))) return rc;
        // in the last block only the CLS row reaches the output (model/make_model.py:98-100): the
        // attention runs the first query tile only and everything after it runs on the B CLS rows.
        ptok = mpreid_prof_begin(stream);
        if (split)
            rc = attention_split_dispatch(reinterpret_cast<const float *>(qkv), B, L, W, cfg->heads, a, tail ? 1 : 0, stream);
        else
            rc = attention_dispatch(qkv, B, L, W, cfg->heads, a, tail ? 1 : 0, stream);
        // algorithmic bytes: k, v of every token + q and the output of the query rows (all of them, or the CLS tile's 16)
        mpreid_prof_end(ptok, stream, MPREID_PROF_ATTENTION, v.M, tail ? 1 : 0, pr,
                        ((double)v.M * 2.0 * W + (tail ? (double)B * 16 : (double)v.M) * 2.0 * W) * 2.0 * pr);
        if (rc) return rc;
        float *xr = x;
        _Float16 *ar = a, *hr = hbuf;
        int rows = v.M, rows_pad = v.Mpad;
        if (tail) {
            hipLaunchKernelGGL(gather_cls_kernel, dim3((unsigned)B), dim3(256), 0, stream, x, a, B, L, W, pr * W, x_cls, a_cls);
            LAUNCH_CHECK();
            xr = x_cls; ar = a_cls; hr = h_cls; rows = B; rows_pad = v.Bpad;
        }
        g = GemmArgs{};
        g.A = ar; g.W = (const _Float16 *)ly.out_proj_w; g.M = rows_pad; g.N = W;
        g.out = xr; g.ldo = W; g.bias = ly.out_proj_b;
        if ((rc = linear(g, W, ly.out_proj_s, GE_BIAS_RES, GE_S_BIAS_RES))) return rc;
        // x = x + c_proj(quickgelu(c_fc(ln_2(x))))
        if (split)
            hipLaunchKernelGGL(layernorm_kernel<2>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, xr,
                               (int64_t)rows, W, ly.ln2_g, ly.ln2_b, (void *)ar, (int64_t)W);
        else
            hipLaunchKernelGGL(layernorm_kernel<1>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, xr,
                               (int64_t)rows, W, ly.ln2_g, ly.ln2_b, (void *)ar, (int64_t)W);
        LAUNCH_CHECK();
        g = GemmArgs{};
        g.A = ar; g.W = (const _Float16 *)ly.fc_w; g.M = rows_pad; g.N = 4 * W;
        g.out = hr; g.ldo = pr * 4 * W; g.bias = ly.fc_b;
        if ((rc = linear(g, W, ly.fc_s, GE_BIAS_GELU, GE_S_BIAS_GELU))) return rc;
        g = GemmArgs{};
        g.A = hr; g.W = (const _Float16 *)ly.proj_w; g.M = rows_pad; g.N = W;
        g.out = xr; g.ldo = W; g.bias = ly.proj_b;
        if ((rc = linear(g, 4 * W, ly.proj_s, GE_BIAS_RES, GE_S_BIAS_RES))) return rc;
    }
    const bool neck = cfg->neck_after != 0 && w->bn_scale && w->bn_proj_scale;
    hipLaunchKernelGGL(cls_ln_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, stream, cls_last ? x_cls : x,
                       cls_last ? (int64_t)W : (int64_t)L * W, B, W, cfg->out_dim, w->ln_post_g, w->ln_post_b,
                       neck ? w->bn_scale : nullptr, neck ? w->bn_shift : nullptr, y_cls, out);
    hipLaunchKernelGGL(cls_proj_kernel, dim3((unsigned)((B + HEAD_IMGS - 1) / HEAD_IMGS), (unsigned)((cfg->out_dim + 255) / 256)),
                       dim3(256), (size_t)HEAD_IMGS * W * 4, stream, y_cls, B, W, cfg->out_dim, w->proj,
                       neck ? w->bn_proj_scale : nullptr, neck ? w->bn_proj_shift : nullptr, out);
    LAUNCH_CHECK();
    return MPREID_OK;
}

// =============================================================================================
// All-fp32 encoder mode (SURVEY.md section 7 hard part 5: "expose an all-fp32 debug mode").
// The fp16-MFMA path above carries a relative feature error of ~4e-4 (operand rounding), which moves mAP by ~1e-4
// on hard data (tests/test_gpu_map_parity.py); this mode keeps every activation and weight in fp32 and runs the
// linear layers on the exact fp32 matrix instruction (gemm_f32_exact_kernel, k-ascending fmaf chains, ~100 TFLOP/s),
// the attention on fp32 vector FMAs.  ~4-5 k images/s: a parity / debugging mode, not the throughput path.
// Same structs as the fp16 entry points, but every *_w / conv_w pointer is an fp32 [out][in] matrix.
// =============================================================================================
int mpreid_gemm_f32_linear(const float *A, const float *Wt, int64_t M, int64_t N, int K, const float *bias, float *C,
                           int64_t ldc, int epi, hipStream_t stream);
enum { F32_LIN = 2, F32_LIN_GELU = 3, F32_LIN_RES = 4 };   // distance.hip: EPI_LIN*

// img [B][3][H][W] fp32 (or img8 [B][H][W][3] uint8 with ToTensor + Normalize applied on the fly, as in im2col_u8_kernel)
// -> patches [B*P][3*p*p] fp32, inner order (c, kh, kw); view = the test-time-augmentation views of im2col_kernel (run-time
// here: the all-fp32 mode is the parity / debugging mode, not a throughput path)
struct F32In {
    const float *img;
    const unsigned char *img8;
    float mean[3], sd[3];
    int view;
};
__device__ __forceinline__ float f32in_px(const F32In &in, int b, int c, int y, int x, int H, int Wd) {
    if (in.img8) return __fdiv_rn(__fdiv_rn((float)in.img8[(((int64_t)b * H + y) * Wd + x) * 3 + c], 255.0f) - in.mean[c], in.sd[c]);
    return in.img[(((int64_t)b * 3 + c) * H + y) * Wd + x];
}
__global__ __launch_bounds__(256) void im2col_f32_kernel(const F32In in, int B, int H, int Wd, int p, int stride,
                                                         int h_res, int w_res, float *__restrict__ out) {
    const int Kp = 3 * p * p, P = h_res * w_res;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (int64_t)B * P * Kp) return;
    const int m = (int)(gid / Kp), k = (int)(gid % Kp);
    const int b = m / P, pi = m % P, ph = pi / w_res, pw = pi % w_res;
    const int c = k / (p * p), kh = (k % (p * p)) / p, kw = k % p;
    const int y = ph * stride + kh, x0 = pw * stride + kw;
    const int x = in.view == 1 ? Wd - 1 - x0 : x0;                       // torch.flip(img, [3])
    float v;
    if (in.view == 2)                                                     // img.mean(dim=1): ((c0 + c1) + c2) / 3
        v = __fdiv_rn((f32in_px(in, b, 0, y, x, H, Wd) + f32in_px(in, b, 1, y, x, H, Wd)) + f32in_px(in, b, 2, y, x, H, Wd), 3.0f);
    else
        v = f32in_px(in, b, in.view == 3 ? 0 : c, y, x, H, Wd);           // 3: img[:, 0:1] in all three channels
    out[gid] = v;
}

// x[b*L + 1 + p][:] = tok[b*P + p][:] + pos[1 + p][:]
__global__ __launch_bounds__(256) void patch_scatter_f32_kernel(const float *__restrict__ tok, const float *__restrict__ pos, int B,
                                                                int P, int L, int W, float *__restrict__ x) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (int64_t)B * P * W) return;
    const int m = (int)(gid / W), k = (int)(gid % W);
    const int b = m / P, pp = m % P;
    x[((int64_t)b * L + 1 + pp) * W + k] = tok[gid] + pos[(int64_t)(1 + pp) * W + k];
}

// softmax(q k^T / sqrt(64)) v per (image, head), fp32 throughout: K and V of the head in LDS, one thread per query
// row, online softmax.  qkv [B*L][3W] (q | k | v column blocks, head h = columns [64h, 64h + 64)), out [B*L][W].
__global__ __launch_bounds__(256) void attention_f32_kernel(const float *__restrict__ qkv, int L, int W, int heads,
                                                            float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *Ks = reinterpret_cast<float *>(smem), *Vs = Ks + (size_t)L * 64;
    const int b = blockIdx.x / heads, h = blockIdx.x % heads, tid = threadIdx.x;
    const float *base = qkv + (int64_t)b * L * 3 * W + h * 64;
    for (int idx = tid; idx < L * 16; idx += 256) {
        const int row = idx >> 4, c4 = (idx & 15) * 4;
        *reinterpret_cast<float4 *>(Ks + row * 64 + c4) = *reinterpret_cast<const float4 *>(base + (int64_t)row * 3 * W + W + c4);
        *reinterpret_cast<float4 *>(Vs + row * 64 + c4) = *reinterpret_cast<const float4 *>(base + (int64_t)row * 3 * W + 2 * W + c4);
    }
    __syncthreads();
    for (int t = tid; t < L; t += 256) {
        float q[64], acc[64];
#pragma unroll
        for (int c = 0; c < 64; c += 4) {
            const float4 v = *reinterpret_cast<const float4 *>(base + (int64_t)t * 3 * W + c);
            q[c] = v.x * 0.125f; q[c + 1] = v.y * 0.125f; q[c + 2] = v.z * 0.125f; q[c + 3] = v.w * 0.125f;   // 64^-0.5, exact
            acc[c] = acc[c + 1] = acc[c + 2] = acc[c + 3] = 0.0f;
        }
        float m = -3.402823466e+38f, l = 0.0f;
        for (int j = 0; j < L; ++j) {
            const float *kr = Ks + j * 64, *vr = Vs + j * 64;
            float s = 0.0f;
#pragma unroll
            for (int c = 0; c < 64; ++c) s = fmaf(q[c], kr[c], s);
            const float mn = fmaxf(m, s);
            const float corr = expf(m - mn), pj = expf(s - mn);
            l = l * corr + pj;
#pragma unroll
            for (int c = 0; c < 64; ++c) acc[c] = fmaf(pj, vr[c], acc[c] * corr);
            m = mn;
        }
        float *o = out + ((int64_t)b * L + t) * W + h * 64;
#pragma unroll
        for (int c = 0; c < 64; ++c) o[c] = __fdiv_rn(acc[c], l);
    }
}

struct VitLayoutF32 {
    int L, P, M, Kp;
    size_t patches, x, a, qkv, hbuf, y_cls, total;
};
static VitLayoutF32 vit_layout_f32(const mpreid_vit_cfg *c, int B) {
    VitLayoutF32 v{};
    v.P = c->h_res * c->w_res;
    v.L = v.P + 1;
    v.M = B * v.L;
    v.Kp = 3 * c->patch * c->patch;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += align_up(bytes, 256);
        return o;
    };
    v.patches = take((size_t)B * v.P * v.Kp * 4);
    v.x = take((size_t)v.M * c->width * 4);
    v.a = take((size_t)v.M * c->width * 4);
    v.qkv = take((size_t)v.M * 3 * c->width * 4);
    v.hbuf = take((size_t)v.M * 4 * c->width * 4);
    v.y_cls = take((size_t)B * c->width * 4);
    v.total = off;
    return v;
}

extern "C" size_t mpreid_vit_workspace_bytes_f32(const mpreid_vit_cfg *cfg, int batch) {
    if (!cfg || batch <= 0) return 0;
    return vit_layout_f32(cfg, batch).total;
}

static int vit_forward_f32_impl(const mpreid_vit_cfg *cfg, const mpreid_vit_weights *w, const F32In &img, int B,
                                const float *cv_emb, float *out, void *ws, size_t ws_bytes, mpreid_stream_t stream_) {
    int rc = vit_check_cfg(cfg);
    if (rc) return rc;
    ARG_CHECK(w && (img.img || img.img8) && img.view >= 0 && img.view <= 3 && out && B > 0 && w->layers);
    if (cfg->width / cfg->heads != 64) {
        mpreid_set_error("fp32 attention: head dimension %d != 64", cfg->width / cfg->heads);
        return MPREID_ERR_UNSUPPORTED;
    }
    const VitLayoutF32 v = vit_layout_f32(cfg, B);
    if (!ws || ws_bytes < v.total) {
        mpreid_set_error("vit fp32 workspace too small: %zu < %zu", ws_bytes, v.total);
        return MPREID_ERR_WORKSPACE;
    }
    hipStream_t stream = (hipStream_t)stream_;
    const int W = cfg->width, L = v.L;
    char *base = (char *)ws;
    float *patches = (float *)(base + v.patches), *x = (float *)(base + v.x), *a = (float *)(base + v.a);
    float *qkv = (float *)(base + v.qkv), *hbuf = (float *)(base + v.hbuf), *y_cls = (float *)(base + v.y_cls);
    {
        const int64_t n = (int64_t)B * v.P * v.Kp;
        hipLaunchKernelGGL(im2col_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, img, B, cfg->img_h,
                           cfg->img_w, cfg->patch, cfg->stride, cfg->h_res, cfg->w_res, patches);
        LAUNCH_CHECK();
        if ((rc = mpreid_gemm_f32_linear(patches, (const float *)w->conv_w, (int64_t)B * v.P, W, v.Kp, nullptr, a, W, F32_LIN,
                                         stream)))
            return rc;
        const int64_t n2 = (int64_t)B * v.P * W;
        hipLaunchKernelGGL(patch_scatter_f32_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, stream, a, w->pos_emb, B,
                           v.P, L, W, x);
        hipLaunchKernelGGL(cls_token_kernel, dim3((unsigned)B), dim3(256), 0, stream, w->class_emb, w->pos_emb, cv_emb, B, L, W,
                           x);
        hipLaunchKernelGGL(layernorm_kernel<0>, dim3((unsigned)((v.M + 3) / 4)), dim3(256), 0, stream, x, (int64_t)v.M, W,
                           w->ln_pre_g, w->ln_pre_b, (void *)x, (int64_t)W);
        LAUNCH_CHECK();
    }
    const size_t att_lds = (size_t)L * 64 * 4 * 2;
    if (att_lds > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(attention_f32_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)att_lds));
    for (int l = 0; l < cfg->layers; ++l) {
        const mpreid_vit_layer &ly = w->layers[l];
        hipLaunchKernelGGL(layernorm_kernel<0>, dim3((unsigned)((v.M + 3) / 4)), dim3(256), 0, stream, x, (int64_t)v.M, W,
                           ly.ln1_g, ly.ln1_b, (void *)a, (int64_t)W);
        LAUNCH_CHECK();
        if ((rc = mpreid_gemm_f32_linear(a, (const float *)ly.in_proj_w, v.M, 3 * W, W, ly.in_proj_b, qkv, 3 * W, F32_LIN, stream)))
            return rc;
        hipLaunchKernelGGL(attention_f32_kernel, dim3((unsigned)(B * cfg->heads)), dim3(256), att_lds, stream, qkv, L, W,
                           cfg->heads, a);
        LAUNCH_CHECK();
        if ((rc = mpreid_gemm_f32_linear(a, (const float *)ly.out_proj_w, v.M, W, W, ly.out_proj_b, x, W, F32_LIN_RES, stream)))
            return rc;
        hipLaunchKernelGGL(layernorm_kernel<0>, dim3((unsigned)((v.M + 3) / 4)), dim3(256), 0, stream, x, (int64_t)v.M, W,
                           ly.ln2_g, ly.ln2_b, (void *)a, (int64_t)W);
        LAUNCH_CHECK();
        if ((rc = mpreid_gemm_f32_linear(a, (const float *)ly.fc_w, v.M, 4 * W, W, ly.fc_b, hbuf, 4 * W, F32_LIN_GELU, stream)))
            return rc;
        if ((rc = mpreid_gemm_f32_linear(hbuf, (const float *)ly.proj_w, v.M, W, 4 * W, ly.proj_b, x, W, F32_LIN_RES, stream)))
            return rc;
    }
    const bool neck = cfg->neck_after != 0 && w->bn_scale && w->bn_proj_scale;
    hipLaunchKernelGGL(cls_ln_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, stream, x, (int64_t)L * W, B, W, cfg->out_dim,
                       w->ln_post_g, w->ln_post_b, neck ? w->bn_scale : nullptr, neck ? w->bn_shift : nullptr, y_cls, out);
    hipLaunchKernelGGL(cls_proj_kernel, dim3((unsigned)((B + HEAD_IMGS - 1) / HEAD_IMGS), (unsigned)((cfg->out_dim + 255) / 256)),
                       dim3(256), (size_t)HEAD_IMGS * W * 4, stream, y_cls, B, W, cfg->out_dim, w->proj,
                       neck ? w->bn_proj_scale : nullptr, neck ? w->bn_proj_shift : nullptr, out);
    LAUNCH_CHECK();
    return MPREID_OK;
}

extern "C" int mpreid_vit_forward_f32(const mpreid_vit_cfg *cfg, const mpreid_vit_weights *w, const float *img, int B,
                                      const float *cv_emb, float *out, void *ws, size_t ws_bytes, mpreid_stream_t stream_) {
    ARG_CHECK(img);
    const F32In in{img, nullptr, {0.f, 0.f, 0.f}, {1.f, 1.f, 1.f}, 0};
    return vit_forward_f32_impl(cfg, w, in, B, cv_emb, out, ws, ws_bytes, stream_);
}

extern "C" int mpreid_vit_forward_f32_view(const mpreid_vit_cfg *cfg, const mpreid_vit_weights *w, const float *img_f32,
                                           const uint8_t *img_u8, const float *mean, const float *stdv, int view, int B,
                                           const float *cv_emb, float *out, void *ws, size_t ws_bytes, mpreid_stream_t stream_) {
    ARG_CHECK((img_f32 != nullptr) != (img_u8 != nullptr) && (!img_u8 || (mean && stdv)));
    F32In in{img_f32, img_u8, {0.f, 0.f, 0.f}, {1.f, 1.f, 1.f}, view};
    if (img_u8)
        for (int c = 0; c < 3; ++c) {
            in.mean[c] = mean[c];
            in.sd[c] = stdv[c];
        }
    return vit_forward_f32_impl(cfg, w, in, B, cv_emb, out, ws, ws_bytes, stream_);
}

extern "C" int mpreid_vit_forward(const mpreid_vit_cfg *cfg, const mpreid_vit_weights *w, const float *img, int B,
                                  const float *cv_emb, float *out, void *ws, size_t ws_bytes, mpreid_stream_t stream) {
    ARG_CHECK(img);
    return vit_forward_impl(cfg, w, img, nullptr, nullptr, nullptr, 0, B, cv_emb, out, ws, ws_bytes, stream);
}

extern "C" int mpreid_vit_forward_u8(const mpreid_vit_cfg *cfg, const mpreid_vit_weights *w, const uint8_t *img_hwc,
                                     const float *pixel_mean3, const float *pixel_std3, int B, const float *cv_emb,
                                     float *out, void *ws, size_t ws_bytes, mpreid_stream_t stream) {
    ARG_CHECK(img_hwc && pixel_mean3 && pixel_std3);
    return vit_forward_impl(cfg, w, nullptr, img_hwc, pixel_mean3, pixel_std3, 0, B, cv_emb, out, ws, ws_bytes, stream);
}

extern "C" int mpreid_vit_forward_view(const mpreid_vit_cfg *cfg, const mpreid_vit_weights *w, const float *img_f32,
                                       const uint8_t *img_hwc_u8, const float *pixel_mean3, const float *pixel_std3,
                                       int view, int B, const float *cv_emb, float *out, void *ws, size_t ws_bytes,
                                       mpreid_stream_t stream) {
    ARG_CHECK((img_f32 != nullptr) != (img_hwc_u8 != nullptr));
    ARG_CHECK(!img_hwc_u8 || (pixel_mean3 && pixel_std3));
    return vit_forward_impl(cfg, w, img_f32, img_hwc_u8, pixel_mean3, pixel_std3, view, B, cv_emb, out, ws, ws_bytes,
                            stream);
}
