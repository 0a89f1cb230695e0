// rn50.hip — CLIP "RN50" image encoder (MODEL.NAME == 'RN50'), evaluation forward, gfx950.
//
// Reference: model/clip/model.py:92-148 (ModifiedResNet), :10-53 (Bottleneck), :56-90 (AttentionPool2d) and the
// RN50 branch of build_transformer.forward (model/make_model.py:82-86, 102-115).
//
// Data layout: activations NHWC fp16 (channels innermost) so that every convolution is an implicit GEMM whose
// A-operand rows are contiguous 128-byte pieces (conv_f16.hip); BatchNorm folded into weights / bias on the host;
// residual add + ReLU in the conv epilogue; AvgPool2d as its own HBM-bound pass.  The 3 -> 32 channel first
// convolution (stride 2, K = 27: no MFMA shape) is a direct fp32 kernel that also does the NCHW fp32 (or uint8 HWC +
// ToTensor + Normalize) -> NHWC fp16 conversion; 32-channel tensors are stored with 64 channels (upper half zero).
//
// Attention pool: only the output at the mean token (index 0) is used by the reference
// (model/make_model.py:86 `image_features_proj[0]`), so: the query projection for token 0 only, the one-query
// attention WITHOUT forming K or V (see "attention pool, one query" below), and c_proj on one row per image.
// avg_pool2d(x4) is the same mean the pool prepends, computed once in fp32.
#include "common.h"
#include "conv_f16.h"
#include "gemm_f16.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// ---- stem conv1 + bn1 + relu: [B][3][H][W] fp32 (or [B][H][W][3] uint8) -> [B][H/2][W/2][64] fp16 -------------
// one thread per output pixel: 27 inputs (coalesced along the image row), all cout <= 32 channels from weights in
// LDS (broadcast float4 reads), 128 contiguous bytes out (channels cout..63 are the zero padding of the storage)
template <bool U8>
__global__ __launch_bounds__(256) void rn50_stem1_kernel(const float *__restrict__ img, const unsigned char *__restrict__ img8,
                                                         float m0, float m1, float m2, float s0, float s1, float s2,
                                                         const float *__restrict__ w, const float *__restrict__ bias,
                                                         int cout, int B, int H, int W, _Float16 *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) float sw[27 * 32 + 32]; // [k = c*9 + kh*3 + kw][n], then bias[n]
    for (int i = threadIdx.x; i < 27 * 32; i += 256) {
        const int k = i >> 5, n = i & 31;
        sw[i] = n < cout ? w[n * 27 + k] : 0.f;
    }
    for (int i = threadIdx.x; i < 32; i += 256) sw[27 * 32 + i] = i < cout ? bias[i] : 0.f;
    __syncthreads();
    const int OH = H / 2, OW = W / 2;
    const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (pix >= (int64_t)B * OH * OW) return;
    const int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH), b = (int)(pix / ((int64_t)OW * OH));
    float x[27]; // [c][kh][kw], zero outside the image (padding = 1)
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int iy = oy * 2 + kh - 1, ix = ox * 2 + kw - 1;
                float v = 0.f;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
                    if (U8) {
                        const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
                        v = __fdiv_rn(__fdiv_rn((float)img8[(((int64_t)b * H + iy) * W + ix) * 3 + c], 255.0f) - mean, sd);
                    } else {
                        v = img[(((int64_t)b * 3 + c) * H + iy) * W + ix];
                    }
                }
                x[c * 9 + kh * 3 + kw] = v;
            }
    _Float16 *o = out + pix * 64;
#pragma unroll
    for (int n8 = 0; n8 < 4; ++n8) {       // 8 output channels at a time: k-ascending fmaf chains from 0, + bias
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 27; ++k) {
            const float4 w0 = *reinterpret_cast<const float4 *>(sw + k * 32 + n8 * 8);
            const float4 w1 = *reinterpret_cast<const float4 *>(sw + k * 32 + n8 * 8 + 4);
            acc[0] = fmaf(x[k], w0.x, acc[0]);
            acc[1] = fmaf(x[k], w0.y, acc[1]);
            acc[2] = fmaf(x[k], w0.z, acc[2]);
            acc[3] = fmaf(x[k], w0.w, acc[3]);
            acc[4] = fmaf(x[k], w1.x, acc[4]);
            acc[5] = fmaf(x[k], w1.y, acc[5]);
            acc[6] = fmaf(x[k], w1.z, acc[6]);
            acc[7] = fmaf(x[k], w1.w, acc[7]);
        }
        h8 hv;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = acc[e] + sw[27 * 32 + n8 * 8 + e];
            hv[e] = (_Float16)(v < 0.f ? 0.f : v);
        }
        *reinterpret_cast<h8 *>(o + n8 * 8) = hv;
    }
    const h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int n8 = 4; n8 < 8; ++n8) *reinterpret_cast<h8 *>(o + n8 * 8) = z;
}

// ---- AvgPool2d(2) on NHWC fp16: ((x00 + x01) + x10 + x11) / 4 in fp32 -----------------------------------------
__global__ __launch_bounds__(256) void avgpool2_kernel(const _Float16 *__restrict__ in, int B, int H, int W, int C,
                                                       _Float16 *__restrict__ out) {
    const int OH = H / 2, OW = W / 2, cc = C / 8;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (int64_t)B * OH * OW * cc) return;
    const int c8 = (int)(gid % cc);
    const int64_t pix = gid / cc;
    const int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH), b = (int)(pix / ((int64_t)OW * OH));
    const _Float16 *p = in + (((int64_t)b * H + oy * 2) * W + ox * 2) * C + c8 * 8;
    const h8 a = *reinterpret_cast<const h8 *>(p), bq = *reinterpret_cast<const h8 *>(p + C);
    const h8 c = *reinterpret_cast<const h8 *>(p + (int64_t)W * C), d = *reinterpret_cast<const h8 *>(p + (int64_t)W * C + C);
    h8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (_Float16)((((float)a[e] + (float)bq[e]) + (float)c[e] + (float)d[e]) * 0.25f);
    *reinterpret_cast<h8 *>(out + pix * C + c8 * 8) = o;
}

// ---- attention-pool tokens (model/clip/model.py:66-69): x4 [B][S][E] fp16 ->
//   mean[b][:] fp32 (= avg_pool2d(x4) = the prepended token), tok[b*(S+1) + 0] = mean + pos[0],
//   tok[b*(S+1) + 1 + t] = x4[b][t] + pos[1 + t], tok0[b] = tok[b*(S+1)]  (fp16 GEMM operands)
__global__ __launch_bounds__(256) void rn50_tokens_kernel(const _Float16 *__restrict__ x4, const float *__restrict__ pos, int S,
                                                          int E, float *__restrict__ mean, _Float16 *__restrict__ tok,
                                                          _Float16 *__restrict__ tok0) {
    const int b = blockIdx.x;
    for (int c8 = threadIdx.x; c8 < E / 8; c8 += 256) {
        float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < S; ++t) {
            const h8 v = *reinterpret_cast<const h8 *>(x4 + ((int64_t)b * S + t) * E + c8 * 8);
            const float *pp = pos + (int64_t)(1 + t) * E + c8 * 8;
            h8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                sum[e] = sum[e] + (float)v[e];
                o[e] = (_Float16)((float)v[e] + pp[e]);
            }
            *reinterpret_cast<h8 *>(tok + ((int64_t)b * (S + 1) + 1 + t) * E + c8 * 8) = o;
        }
        h8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float m = __fdiv_rn(sum[e], (float)S);
            mean[(int64_t)b * E + c8 * 8 + e] = m;
            o[e] = (_Float16)(m + pos[c8 * 8 + e]);
        }
        *reinterpret_cast<h8 *>(tok + (int64_t)b * (S + 1) * E + c8 * 8) = o;
        *reinterpret_cast<h8 *>(tok0 + (int64_t)b * E + c8 * 8) = o;
    }
}

// ---- attention pool, one query (model/clip/model.py:56-90; only the output at the mean token is used) ----------
// The reference projects K and V for all T = S + 1 tokens (2 x T x E x E MACs per image: 2.16 GFLOP at E = 2048, a
// fifth of the tower) to use them against ONE query.  With q = Wq tok_0 + bq per head h (64 dims):
//     score_t = q_h . (Wk_h tok_t + bk_h) / 8 = (Wk_h^T q_h) . tok_t / 8 + const      (the constant cancels in softmax)
//     out_h   = sum_t p_t (Wv_h tok_t + bv_h) = Wv_h (sum_t p_t tok_t) + bv_h
// so K and V are never formed: u_h = Wk_h^T q_h and z_h = sum_t p_t tok_t are E-vectors per (image, head).
//   rn50_pool_expand_kernel : A[b*H + h][:] = q[b] restricted to head h's 64 dims (zeros elsewhere), so that ONE plain
//                             GEMM against Wk^T yields all u_h (32x redundant MACs, still 70 us instead of 540)
//   rn50_pool_core_kernel   : per image: scores of the T tokens against the H vectors u_h (in LDS), softmax, z_h
//   second GEMM             : z_h against Wv (+ bv); rn50_pool_gather_kernel keeps head h's own 64 outputs
__global__ __launch_bounds__(256) void rn50_pool_expand_kernel(const _Float16 *__restrict__ q, int H, int E,
                                                               _Float16 *__restrict__ a_exp) {
    const int64_t row = blockIdx.x;   // b * H + h
    const int b = (int)(row / H), h = (int)(row % H);
    const int hd = E / H;
    for (int c8 = threadIdx.x; c8 < E / 8; c8 += 256) {
        h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if ((c8 * 8) / hd == h) v = *reinterpret_cast<const h8 *>(q + (int64_t)b * E + c8 * 8);
        *reinterpret_cast<h8 *>(a_exp + row * E + c8 * 8) = v;
    }
}

typedef _Float16 pf16x8 __attribute__((ext_vector_type(8)));
typedef float pf32x4 __attribute__((ext_vector_type(4)));
typedef float pf32x2 __attribute__((ext_vector_type(2)));

// One workgroup (4 waves) per image.
//   scores  S^T[t][h] = tok_t . u_h on the matrix cores: the token rows are the MFMA A operand exactly as they lie in
//           memory (16 B per lane straight from global, eight k-steps requested ahead), the u rows the B operand from
//           LDS (row pitch E + 8 halfs: the 16 rows of a fragment fall on distinct banks); wave w takes token tiles
//           w, w + 4, ...
//   softmax one thread per head over its T scores
//   z       z_h = sum_t p[h][t] tok_t: the contraction runs over the SLOW index of tok, which the MFMA operand
//           layout cannot read without a transpose, so this part is packed fp32 FMAs: a thread owns 8 columns and 8
//           heads at a time, token rows prefetched four ahead
__global__ __launch_bounds__(512) void rn50_pool_core_kernel(const _Float16 *__restrict__ U, const _Float16 *__restrict__ tok,
                                                             int T, int E, int H, _Float16 *__restrict__ Z) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int ES = E + 8;
    const int H16 = (H + 15) & ~15;
    _Float16 *Us = reinterpret_cast<_Float16 *>(lds);                  // [H16][ES], rows >= H zero
    float *sc = reinterpret_cast<float *>(lds + (size_t)H16 * ES * 2);  // [T][H16] (token-major: a token's H weights are one vector)
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const _Float16 *Ub = U + (int64_t)b * H * E;
    const _Float16 *tb = tok + (int64_t)b * T * E;
    for (int i = tid; i < H16 * (E / 8); i += 512) {
        const int h = i / (E / 8), c8 = i % (E / 8);
        pf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (h < H) v = *reinterpret_cast<const pf16x8 *>(Ub + (int64_t)h * E + c8 * 8);
        *reinterpret_cast<pf16x8 *>(Us + h * ES + c8 * 8) = v;
    }
    __syncthreads();
    {
        const float scale = 1.0f / __fsqrt_rn((float)(E / H));
        const int frow = lane & 15, fq = lane >> 4;
        const int nsteps = E / 32, ntt = (T + 15) / 16, nht = H16 / 16;
        for (int tt = wave; tt < ntt; tt += 8) {
            const int t = tt * 16 + frow;
            const _Float16 *ar = tb + (int64_t)(t < T ? t : T - 1) * E + fq * 8;
            const _Float16 *br = Us + frow * ES + fq * 8;
            pf32x4 acc[2] = {pf32x4{0.f, 0.f, 0.f, 0.f}, pf32x4{0.f, 0.f, 0.f, 0.f}};
            constexpr int PD = 8;   // k-steps of A fragments in flight
            pf16x8 an[PD];
#pragma unroll
            for (int u = 0; u < PD; ++u) an[u] = *reinterpret_cast<const pf16x8 *>(ar + (u < nsteps ? u : 0) * 32);
            for (int k0 = 0; k0 < nsteps; k0 += PD) {
                pf16x8 ac[PD];
#pragma unroll
                for (int u = 0; u < PD; ++u) ac[u] = an[u];
#pragma unroll
                for (int u = 0; u < PD; ++u) {
                    const int kn = k0 + PD + u;
                    an[u] = *reinterpret_cast<const pf16x8 *>(ar + (kn < nsteps ? kn : 0) * 32);
                }
#pragma unroll
                for (int u = 0; u < PD; ++u) {
                    if (k0 + u < nsteps) {
                        const pf16x8 b0 = *reinterpret_cast<const pf16x8 *>(br + (k0 + u) * 32);
                        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ac[u], b0, acc[0], 0, 0, 0);
                        if (nht > 1) {
                            const pf16x8 b1 = *reinterpret_cast<const pf16x8 *>(br + 16 * ES + (k0 + u) * 32);
                            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ac[u], b1, acc[1], 0, 0, 0);
                        }
                    }
                }
            }
            // C layout: col = lane & 15 (head), row = (lane >> 4) * 4 + r (token)
#pragma unroll
            for (int ht = 0; ht < 2; ++ht) {
                if (ht >= nht) break;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int tr = tt * 16 + fq * 4 + r;
                    if (tr < T) sc[tr * H16 + ht * 16 + frow] = acc[ht][r] * scale;
                }
            }
        }
    }
    __syncthreads();
    // softmax over the T tokens of each head: a wave per head (lanes over the tokens); padding heads get zero weights
    for (int h = wave; h < H16; h += 8) {
        float *r = sc + h;
        if (h >= H) {
            for (int t = lane; t < T; t += 64) r[t * H16] = 0.f;
            continue;
        }
        float mx = -3.0e38f;
        for (int t = lane; t < T; t += 64) mx = fmaxf(mx, r[t * H16]);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        float sum = 0.f;
        for (int t = lane; t < T; t += 64) {
            const float p = __expf(r[t * H16] - mx);
            r[t * H16] = p;
            sum += p;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
        const float inv = __fdiv_rn(1.0f, sum);
        for (int t = lane; t < T; t += 64) r[t * H16] *= inv;
    }
    __syncthreads();
    // z: a thread owns 4 columns and up to 32 heads (one pass over the token rows: they are the HBM traffic here)
    typedef _Float16 pf16x4 __attribute__((ext_vector_type(4)));
    for (int c4 = tid; c4 < E / 4; c4 += 512) {
        const _Float16 *tc = tb + c4 * 4;
        for (int h0 = 0; h0 < H16; h0 += 32) {
            const int nh = (H16 - h0 < 32) ? H16 - h0 : 32;   // 16 or 32
            pf32x2 acc[32][2];
#pragma unroll
            for (int hh = 0; hh < 32; ++hh) acc[hh][0] = acc[hh][1] = pf32x2{0.f, 0.f};
            constexpr int PD = 8;
            pf16x4 tn[PD];
#pragma unroll
            for (int u = 0; u < PD; ++u) tn[u] = *reinterpret_cast<const pf16x4 *>(tc + (int64_t)(u < T ? u : 0) * E);
            for (int t0 = 0; t0 < T; t0 += PD) {
                pf16x4 tcur[PD];
#pragma unroll
                for (int u = 0; u < PD; ++u) tcur[u] = tn[u];
#pragma unroll
                for (int u = 0; u < PD; ++u) {
                    const int tnx = t0 + PD + u;
                    tn[u] = *reinterpret_cast<const pf16x4 *>(tc + (int64_t)(tnx < T ? tnx : 0) * E);
                }
#pragma unroll
                for (int u = 0; u < PD; ++u) {
                    const int t = t0 + u;
                    if (t < T) {
                        const pf32x2 tv0 = {(float)tcur[u][0], (float)tcur[u][1]}, tv1 = {(float)tcur[u][2], (float)tcur[u][3]};
                        const float *pr = sc + t * H16 + h0;
#pragma unroll
                        for (int h4 = 0; h4 < 8; ++h4) {
                            if (h4 * 4 < nh) {
                                const pf32x4 p4 = *reinterpret_cast<const pf32x4 *>(pr + h4 * 4);
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    const pf32x2 p2 = {p4[q], p4[q]};
                                    acc[h4 * 4 + q][0] = __builtin_elementwise_fma(p2, tv0, acc[h4 * 4 + q][0]);
                                    acc[h4 * 4 + q][1] = __builtin_elementwise_fma(p2, tv1, acc[h4 * 4 + q][1]);
                                }
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int hh = 0; hh < 32; ++hh) {
                if (h0 + hh < H) {
                    const pf16x4 o = {(_Float16)acc[hh][0][0], (_Float16)acc[hh][0][1], (_Float16)acc[hh][1][0],
                                      (_Float16)acc[hh][1][1]};
                    *reinterpret_cast<pf16x4 *>(Z + ((int64_t)b * H + h0 + hh) * E + c4 * 4) = o;
                }
            }
        }
    }
}

// att[b][h*hd + d] = O[(b*H + h)][h*hd + d]: head h keeps its own output dims of the redundant product
__global__ __launch_bounds__(256) void rn50_pool_gather_kernel(const _Float16 *__restrict__ O, int H, int E,
                                                               _Float16 *__restrict__ att) {
    const int b = blockIdx.x, hd = E / H;
    for (int c8 = threadIdx.x; c8 < E / 8; c8 += 256) {
        const int h = (c8 * 8) / hd;
        *reinterpret_cast<h8 *>(att + (int64_t)b * E + c8 * 8) =
            *reinterpret_cast<const h8 *>(O + ((int64_t)b * H + h) * E + c8 * 8);
    }
}

// ---- head: out[b] = cat(mean[b] (E), proj[b][:out_dim]) (* scale + shift for NECK_FEAT == 'after') ----------
__global__ __launch_bounds__(256) void rn50_head_kernel(const float *__restrict__ mean, const float *__restrict__ proj, int E,
                                                        int out_dim, int ldp, const float *__restrict__ scale,
                                                        const float *__restrict__ shift, float *__restrict__ out) {
    const int b = blockIdx.x, D = E + out_dim;
    for (int c = threadIdx.x; c < D; c += 256) {
        float v = c < E ? mean[(int64_t)b * E + c] : proj[(int64_t)b * ldp + (c - E)];
        if (scale) v = fmaf(v, scale[c], shift[c]);
        out[(int64_t)b * D + c] = v;
    }
}

struct Rn50Layout {
    int S, T, E, out_pad;
    int64_t tok_rows, b_pad, q_rows;
    size_t act_elems;   // elements of one activation buffer
    size_t zero, act[5], mean, tok, tok0, aexp, uvec, zvec, ofull, zbias, q, att, proj, total;
};

Rn50Layout rn50_layout(const mpreid_rn50_cfg *cfg, int B) {
    Rn50Layout v{};
    const int fh = cfg->img_h / 16, fw = cfg->img_w / 16;
    v.S = fh * fw;
    v.T = v.S + 1;
    v.E = cfg->width * 32;
    v.out_pad = (int)align_up((size_t)cfg->out_dim, 128);
    v.tok_rows = (int64_t)align_up((size_t)B * v.T, 256);
    v.b_pad = (int64_t)align_up((size_t)B, 256);
    v.q_rows = (int64_t)align_up((size_t)B * cfg->heads, 256);
    // the largest tensors: stem outputs [H/2][W/2][64] and layer1 outputs [H/4][W/4][4*width]
    const size_t stem = (size_t)(cfg->img_h / 2) * (cfg->img_w / 2) * 64;
    const size_t l1 = (size_t)(cfg->img_h / 4) * (cfg->img_w / 4) * (size_t)(cfg->width * 4);
    v.act_elems = (size_t)B * (stem > l1 ? stem : l1) + 128 * 64;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off += align_up(bytes, 256);
        return o;
    };
    v.zero = take(256);
    for (int i = 0; i < 5; ++i) v.act[i] = take(v.act_elems * 2);
    v.mean = take((size_t)B * v.E * 4);
    v.tok = take((size_t)v.tok_rows * v.E * 2);
    v.tok0 = take((size_t)v.b_pad * v.E * 2);
    v.aexp = take((size_t)v.q_rows * v.E * 2);
    v.uvec = take((size_t)v.q_rows * v.E * 2);
    v.zvec = take((size_t)v.q_rows * v.E * 2);
    v.ofull = take((size_t)v.q_rows * v.E * 2);
    v.zbias = take((size_t)v.E * 4);
    v.q = take((size_t)v.b_pad * v.E * 2);
    v.att = take((size_t)v.b_pad * v.E * 2);
    v.proj = take((size_t)v.b_pad * v.out_pad * 4);
    v.total = off;
    return v;
}

int rn50_check_cfg(const mpreid_rn50_cfg *c) {
    ARG_CHECK(c != nullptr);
    ARG_CHECK(c->img_h > 0 && c->img_w > 0 && c->img_h % 32 == 0 && c->img_w % 32 == 0);
    ARG_CHECK(c->width >= 16 && c->width % 16 == 0 && c->width <= 64 && c->n_blocks >= 4);
    ARG_CHECK(c->heads > 0 && (c->width * 32) % c->heads == 0 && (c->width * 32) / c->heads == 64);
    ARG_CHECK(c->out_dim > 0 && c->out_dim % 8 == 0 && (c->img_h / 16) * (c->img_w / 16) + 1 <= 256);
    return 0;
}

int run_conv(const mpreid_rn50_conv &c, const _Float16 *in, int B, int H, int W, const _Float16 *identity, int relu,
             _Float16 *out, const _Float16 *zero, hipStream_t stream) {
    const int64_t M = (int64_t)B * H * W;
    if (c.taps == 1 && c.cout % 128 == 0 && c.cout == c.cout_pad && M % 128 == 0 && (relu || !identity)) {
        // a 1x1 convolution over NHWC is a plain GEMM [M][cin] x [cout][cin]^T: the fp16 GEMM kernels of the ViT
        // path (persistent 256x256 tiles when the grid fills the chip) run it 25-45 % faster than the implicit-conv
        // kernel (tools/conv_vs_gemm.py); their ReLU / residual epilogues do the same arithmetic, bit for bit
        GemmArgs g{};
        g.A = in;
        g.W = (const _Float16 *)c.w;
        g.M = (int)M;
        g.N = c.cout;
        g.K = c.cin;
        g.out = out;
        g.ldo = c.cout;
        g.bias = c.bias;
        g.identity = identity;
        return launch_gemm_f16(g, identity ? GE_BIAS_ADD_RELU : (relu ? GE_BIAS_RELU : GE_BIAS_F16), stream);
    }
    ConvArgs a{};
    a.act = in;
    a.wgt = (const _Float16 *)c.w;
    a.bias = c.bias;
    a.identity = identity;
    a.out = out;
    a.zero_page = zero;
    a.H = H;
    a.W = W;
    a.C = c.cin;
    a.M = B * H * W;
    a.N = c.cout;
    a.Npad = c.cout_pad;
    a.ldo = c.cout;
    a.taps = c.taps;
    a.relu = relu;
    return launch_conv_f16(a, stream);
}

int run_pool(const _Float16 *in, int B, int H, int W, int C, _Float16 *out, hipStream_t stream) {
    const int64_t threads = (int64_t)B * (H / 2) * (W / 2) * (C / 8);
    hipLaunchKernelGGL(avgpool2_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, in, B, H, W, C, out);
    LAUNCH_CHECK();
    return 0;
}

} // namespace

extern "C" size_t mpreid_rn50_workspace_bytes(const mpreid_rn50_cfg *cfg, int batch) {
    if (rn50_check_cfg(cfg) || batch <= 0) return 0;
    return rn50_layout(cfg, batch).total;
}

extern "C" int mpreid_rn50_forward(const mpreid_rn50_cfg *cfg, const mpreid_rn50_weights *w, const float *img,
                                   const uint8_t *img8, const float *mean3, const float *std3, int B, float *out,
                                   void *ws, size_t ws_bytes, mpreid_stream_t stream_) {
    int rc = rn50_check_cfg(cfg);
    if (rc) return rc;
    ARG_CHECK(w && out && B > 0 && w->blocks && w->stem1_w && w->stem1_b && w->kt_w && w->v_w && w->v_b && w->q_w);
    ARG_CHECK((img != nullptr) != (img8 != nullptr));
    ARG_CHECK(!img8 || (mean3 && std3));
    const Rn50Layout v = rn50_layout(cfg, B);
    if (!ws || ws_bytes < v.total) {
        mpreid_set_error("rn50 workspace too small: %zu < %zu", ws_bytes, v.total);
        return MPREID_ERR_WORKSPACE;
    }
    hipStream_t stream = (hipStream_t)stream_;
    char *base = (char *)ws;
    const _Float16 *zero = (const _Float16 *)(base + v.zero);
    _Float16 *buf[5];
    for (int i = 0; i < 5; ++i) buf[i] = (_Float16 *)(base + v.act[i]);
    HIP_TRY(hipMemsetAsync(base + v.zero, 0, 256, stream));

    // ---- stem (model/clip/model.py:128-134): conv1/2/3 + bn + relu, AvgPool2d(2) ----
    int H = cfg->img_h / 2, W = cfg->img_w / 2;
    {
        const int64_t threads = (int64_t)B * H * W;   // one per output pixel
        const dim3 grid((unsigned)((threads + 255) / 256));
        if (img8)
            hipLaunchKernelGGL(rn50_stem1_kernel<true>, grid, dim3(256), 0, stream, nullptr, img8, mean3[0], mean3[1],
                               mean3[2], std3[0], std3[1], std3[2], w->stem1_w, w->stem1_b, cfg->width / 2, B, cfg->img_h,
                               cfg->img_w, buf[0]);
        else
            hipLaunchKernelGGL(rn50_stem1_kernel<false>, grid, dim3(256), 0, stream, img, nullptr, 0.f, 0.f, 0.f, 1.f, 1.f,
                               1.f, w->stem1_w, w->stem1_b, cfg->width / 2, B, cfg->img_h, cfg->img_w, buf[0]);
        LAUNCH_CHECK();
    }
    if ((rc = run_conv(w->stem2, buf[0], B, H, W, nullptr, 1, buf[1], zero, stream))) return rc;
    if ((rc = run_conv(w->stem3, buf[1], B, H, W, nullptr, 1, buf[0], zero, stream))) return rc;
    if ((rc = run_pool(buf[0], B, H, W, w->stem3.cout, buf[1], stream))) return rc;
    H /= 2;
    W /= 2;
    int xi = 1; // index of the buffer that holds the block input x

    // ---- residual layers (model/clip/model.py:39-53) ----
    for (int bi = 0; bi < cfg->n_blocks; ++bi) {
        const mpreid_rn50_block &blk = w->blocks[bi];
        ARG_CHECK(blk.stride == 1 || blk.stride == 2);
        int free_i[4], nf = 0;
        for (int i = 0; i < 5; ++i)
            if (i != xi) free_i[nf++] = i;
        _Float16 *x = buf[xi], *t1 = buf[free_i[0]], *t2 = buf[free_i[1]], *t3 = buf[free_i[2]], *t4 = buf[free_i[3]];
        if ((rc = run_conv(blk.conv1, x, B, H, W, nullptr, 1, t1, zero, stream))) return rc;
        if ((rc = run_conv(blk.conv2, t1, B, H, W, nullptr, 1, t2, zero, stream))) return rc;
        int OH = H, OW = W;
        const _Float16 *o2 = t2;
        if (blk.stride == 2) {
            if ((rc = run_pool(t2, B, H, W, blk.conv2.cout, t1, stream))) return rc;
            o2 = t1;
            OH = H / 2;
            OW = W / 2;
        }
        const _Float16 *idt = x;
        if (blk.down.w) {
            const _Float16 *xin = x;
            if (blk.stride == 2) {
                if ((rc = run_pool(x, B, H, W, blk.down.cin, t3, stream))) return rc;
                xin = t3;
            }
            if ((rc = run_conv(blk.down, xin, B, OH, OW, nullptr, 0, t4, zero, stream))) return rc;
            idt = t4;
        } else {
            ARG_CHECK(blk.stride == 1 && blk.conv3.cout == blk.conv1.cin);
        }
        // out = relu(bn3(conv3(o2)) + identity); written to a buffer that is neither o2 nor the identity
        _Float16 *dst = (o2 == t1) ? t2 : t1;
        if ((rc = run_conv(blk.conv3, o2, B, OH, OW, idt, 1, dst, zero, stream))) return rc;
        xi = (dst == t1) ? free_i[0] : free_i[1];
        H = OH;
        W = OW;
    }
    ARG_CHECK(H * W == v.S && w->blocks[cfg->n_blocks - 1].conv3.cout == v.E);

    // ---- attention pool + head ----
    float *mean = (float *)(base + v.mean);
    _Float16 *tok = (_Float16 *)(base + v.tok), *tok0 = (_Float16 *)(base + v.tok0);
    _Float16 *q = (_Float16 *)(base + v.q), *att = (_Float16 *)(base + v.att);
    _Float16 *aexp = (_Float16 *)(base + v.aexp), *uvec = (_Float16 *)(base + v.uvec), *zvec = (_Float16 *)(base + v.zvec);
    _Float16 *ofull = (_Float16 *)(base + v.ofull);
    float *zbias = (float *)(base + v.zbias);
    float *proj = (float *)(base + v.proj);
    const int Hh = cfg->heads;
    hipLaunchKernelGGL(rn50_tokens_kernel, dim3(B), dim3(256), 0, stream, buf[xi], w->pos_emb, v.S, v.E, mean, tok, tok0);
    LAUNCH_CHECK();
    HIP_TRY(hipMemsetAsync(zbias, 0, (size_t)v.E * 4, stream));
    {
        GemmArgs g{};   // q = Wq tok_0 + bq
        g.A = tok0;
        g.W = (const _Float16 *)w->q_w;
        g.M = (int)v.b_pad;
        g.N = v.E;
        g.K = v.E;
        g.out = q;
        g.ldo = v.E;
        g.bias = w->q_b;
        if ((rc = launch_gemm_f16(g, GE_BIAS_F16, stream))) return rc;
        hipLaunchKernelGGL(rn50_pool_expand_kernel, dim3((unsigned)(B * Hh)), dim3(256), 0, stream, q, Hh, v.E, aexp);
        LAUNCH_CHECK();
        g.A = aexp;     // u_h = Wk_h^T q_h for every (image, head)
        g.W = (const _Float16 *)w->kt_w;
        g.M = (int)v.q_rows;
        g.out = uvec;
        g.bias = zbias;
        if ((rc = launch_gemm_f16(g, GE_BIAS_F16, stream))) return rc;
    }
    {
        const int H16 = (Hh + 15) & ~15;
        const size_t lds = (size_t)H16 * (v.E + 8) * 2 + (size_t)H16 * v.T * 4;
        ARG_CHECK(lds <= 160 * 1024 && v.E % 32 == 0 && H16 % 4 == 0);
        static size_t lds_set = 0;
        if (lds > 48 * 1024 && lds > lds_set) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(rn50_pool_core_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            lds_set = lds;
        }
        hipLaunchKernelGGL(rn50_pool_core_kernel, dim3(B), dim3(512), lds, stream, uvec, tok, v.T, v.E, Hh, zvec);
        LAUNCH_CHECK();
        GemmArgs g{};   // out_h = Wv_h z_h + bv_h (all heads' rows against all of Wv; the gather keeps the diagonal blocks)
        g.A = zvec;
        g.W = (const _Float16 *)w->v_w;
        g.M = (int)v.q_rows;
        g.N = v.E;
        g.K = v.E;
        g.out = ofull;
        g.ldo = v.E;
        g.bias = w->v_b;
        if ((rc = launch_gemm_f16(g, GE_BIAS_F16, stream))) return rc;
        hipLaunchKernelGGL(rn50_pool_gather_kernel, dim3(B), dim3(256), 0, stream, ofull, Hh, v.E, att);
        LAUNCH_CHECK();
    }
    {
        HIP_TRY(hipMemsetAsync(proj, 0, (size_t)v.b_pad * v.out_pad * 4, stream));
        GemmArgs g{};
        g.A = att;
        g.W = (const _Float16 *)w->c_w;
        g.M = (int)v.b_pad;
        g.N = v.out_pad;
        g.K = v.E;
        g.out = proj;
        g.ldo = v.out_pad;
        g.bias = w->c_b;
        if ((rc = launch_gemm_f16(g, GE_BIAS_RES, stream))) return rc;
    }
    hipLaunchKernelGGL(rn50_head_kernel, dim3(B), dim3(256), 0, stream, mean, proj, v.E, cfg->out_dim, v.out_pad,
                       w->bn_scale, w->bn_shift, out);
    LAUNCH_CHECK();
    return 0;
}
