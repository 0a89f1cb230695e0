// rn50_f32.hip — all-fp32 mode of the CLIP "RN50" image encoder (MODEL.NAME == 'RN50', MODEL.ENCODER_PRECISION: fp32).
//
// Reference: model/clip/model.py:92-148 (ModifiedResNet), :10-53 (Bottleneck), :56-90 (AttentionPool2d) and the RN50
// branch of build_transformer.forward (model/make_model.py:82-86, 102-115) -- the reference runs them in fp32.
//
// The fp16 tower (rn50.hip: fp16 activations AND an fp16 residual stream) carries a relative feature error of 2.6e-3:
// good for throughput, 26x too coarse for the 1e-4 mAP bound.  This mode keeps every activation and weight in fp32 and
// runs every convolution as a GEMM on the EXACT fp32 matrix instruction (gemm_f32_exact_kernel: k-ascending fmaf chains):
//   1x1 convolutions   NHWC activations [B*H*W][Cin] x W[Cout][Cin]^T directly
//   3x3 convolutions   im2col (pad 1, stride 1) -> [B*H*W][9*Cin], k order (kh, kw, c), x W[Cout][9*Cin]^T
// with BatchNorm folded into weights / bias on the host (fp64, rounded once), ReLU and the residual add in the GEMM
// epilogue (EPI_LIN_RELU / EPI_LIN_RES_RELU: the identity sits in the output buffer), AvgPool2d and the attention pool
// (explicit q / k / v projections, one query per image: token 0 is the only output the reference uses) in fp32.
// A parity / debugging mode (~1/10 of the fp16 tower's throughput), the RN50 counterpart of mpreid_vit_forward_f32.
#include "common.h"

int mpreid_gemm_f32_linear(const float *A, const float *Wt, int64_t M, int64_t N, int K, const float *bias, float *C,
                           int64_t ldc, int epi, hipStream_t stream);
enum { F32_LIN = 2, F32_LIN_RES = 4, F32_LIN_RELU = 5, F32_LIN_RES_RELU = 6 };   // distance.hip: EPI_LIN*

namespace {

// stem conv1 + bn1 + relu: [B][3][H][W] fp32 -> [B][H/2][W/2][cout] fp32 (stride 2, pad 1); w [cout][c][kh][kw] folded
__global__ __launch_bounds__(256) void stem1_f32_kernel(const float *__restrict__ img, const float *__restrict__ w,
                                                        const float *__restrict__ bias, int cout, int B, int H, int W,
                                                        float *__restrict__ out) {
    const int OH = H / 2, OW = W / 2;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (int64_t)B * OH * OW * cout) return;
    const int n = (int)(gid % cout);
    const int64_t pix = gid / cout;
    const int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH), b = (int)(pix / ((int64_t)OW * OH));
    float acc = 0.0f;
    for (int c = 0; c < 3; ++c)
        for (int kh = 0; kh < 3; ++kh)
            for (int kw = 0; kw < 3; ++kw) {
                const int iy = oy * 2 + kh - 1, ix = ox * 2 + kw - 1;
                const float x = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? img[(((int64_t)b * 3 + c) * H + iy) * W + ix] : 0.0f;
                acc = fmaf(x, w[n * 27 + c * 9 + kh * 3 + kw], acc);
            }
    const float v = acc + bias[n];
    out[gid] = v < 0.0f ? 0.0f : v;
}

// [B][H][W][C] -> [B*H*W][9*C], k = (kh*3 + kw)*C + c, zero padded (pad 1, stride 1); C % 4 == 0
__global__ __launch_bounds__(256) void im2col3x3_f32_kernel(const float *__restrict__ in, int B, int H, int W, int C,
                                                            float *__restrict__ out) {
    const int c4n = C / 4;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (int64_t)B * H * W * 9 * c4n) return;
    const int c4 = (int)(gid % c4n);
    const int tap = (int)((gid / c4n) % 9);
    const int64_t pix = gid / ((int64_t)c4n * 9);
    const int x = (int)(pix % W), y = (int)((pix / W) % H), b = (int)(pix / ((int64_t)W * H));
    const int iy = y + tap / 3 - 1, ix = x + tap % 3 - 1;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (iy >= 0 && iy < H && ix >= 0 && ix < W)
        v = *reinterpret_cast<const float4 *>(in + (((int64_t)b * H + iy) * W + ix) * C + c4 * 4);
    *reinterpret_cast<float4 *>(out + pix * 9 * C + (int64_t)tap * C + c4 * 4) = v;
}

// AvgPool2d(2) on NHWC fp32
__global__ __launch_bounds__(256) void avgpool2_f32_kernel(const float *__restrict__ in, int B, int H, int W, int C,
                                                           float *__restrict__ out) {
    const int OH = H / 2, OW = W / 2;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (int64_t)B * OH * OW * C) return;
    const int c = (int)(gid % C);
    const int64_t pix = gid / C;
    const int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH), b = (int)(pix / ((int64_t)OW * OH));
    const float *p = in + (((int64_t)b * H + oy * 2) * W + ox * 2) * C + c;
    out[gid] = (((p[0] + p[C]) + p[(int64_t)W * C]) + p[(int64_t)W * C + C]) * 0.25f;
}

// attention-pool tokens (model/clip/model.py:66-69): x4 [B][S][E] -> mean[b] (= avg_pool2d(x4), the prepended token),
// tok[b][0] = mean + pos[0], tok[b][1 + t] = x4[b][t] + pos[1 + t]
__global__ __launch_bounds__(256) void tokens_f32_kernel(const float *__restrict__ x4, const float *__restrict__ pos, int S, int E,
                                                         float *__restrict__ mean, float *__restrict__ tok) {
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < E; c += 256) {
        float sum = 0.0f;
        for (int t = 0; t < S; ++t) {
            const float v = x4[((int64_t)b * S + t) * E + c];
            sum += v;
            tok[((int64_t)b * (S + 1) + 1 + t) * E + c] = v + pos[(int64_t)(1 + t) * E + c];
        }
        const float m = sum / (float)S;
        mean[(int64_t)b * E + c] = m;
        tok[(int64_t)b * (S + 1) * E + c] = m + pos[c];
    }
}

// one-query attention per (image, head): q = q0[b][h*64 ..] * 64^-0.5, scores over the T tokens' k, softmax, weighted v.
// k, v [B*T][E]; q0 [B][E] (the projected token 0); out [B][E].  One 64-thread wave per (image, head), head dim 64.
__global__ __launch_bounds__(64) void pool_attend_f32_kernel(const float *__restrict__ q0, const float *__restrict__ k,
                                                             const float *__restrict__ v, int T, int E, int heads,
                                                             float *__restrict__ out) {
    extern __shared__ float sc[];   // [T]
    const int b = blockIdx.x / heads, h = blockIdx.x % heads, lane = threadIdx.x;
    const float qd = q0[(int64_t)b * E + h * 64 + lane] * 0.125f;   // head_dim ** -0.5, exact
    for (int t = 0; t < T; ++t) {
        float p = qd * k[((int64_t)b * T + t) * E + h * 64 + lane];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) p += __shfl_xor(p, off, 64);
        if (lane == 0) sc[t] = p;
    }
    __syncthreads();
    float mx = -3.402823466e+38f;
    for (int t = 0; t < T; ++t) mx = fmaxf(mx, sc[t]);
    float sum = 0.0f, acc = 0.0f;
    for (int t = 0; t < T; ++t) {
        const float e = expf(sc[t] - mx);
        sum += e;
        acc = fmaf(e, v[((int64_t)b * T + t) * E + h * 64 + lane], acc);
    }
    out[(int64_t)b * E + h * 64 + lane] = __fdiv_rn(acc, sum);
}

// rows 0 of every image's token block -> compact [B][E]
__global__ __launch_bounds__(256) void gather_tok0_f32_kernel(const float *__restrict__ tok, int T, int E, float *__restrict__ tok0) {
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < E; c += 256) tok0[(int64_t)b * E + c] = tok[(int64_t)b * T * E + c];
}

// head: out[b] = cat(mean[b] (E), proj[b] (out_dim)) (* scale + shift for NECK_FEAT == 'after')
__global__ __launch_bounds__(256) void head_f32_kernel(const float *__restrict__ mean, const float *__restrict__ proj, int E,
                                                       int out_dim, const float *__restrict__ scale,
                                                       const float *__restrict__ shift, float *__restrict__ out) {
    const int b = blockIdx.x, D = E + out_dim;
    for (int c = threadIdx.x; c < D; c += 256) {
        float v = c < E ? mean[(int64_t)b * E + c] : proj[(int64_t)b * out_dim + (c - E)];
        if (scale) v = fmaf(v, scale[c], shift[c]);
        out[(int64_t)b * D + c] = v;
    }
}

struct LayoutF32 {
    int S, T, E;
    size_t act_elems, col_elems;
    size_t act[5], col, mean, tok, tok0, q, k, v, att, proj, total;
};

LayoutF32 layout_f32(const mpreid_rn50_cfg *cfg, int B) {
    LayoutF32 v{};
    const int fh = cfg->img_h / 16, fw = cfg->img_w / 16;
    v.S = fh * fw;
    v.T = v.S + 1;
    v.E = cfg->width * 32;
    const size_t stem = (size_t)(cfg->img_h / 2) * (cfg->img_w / 2) * (size_t)cfg->width;       // stem3 output
    const size_t l1 = (size_t)(cfg->img_h / 4) * (cfg->img_w / 4) * (size_t)(cfg->width * 4);     // layer1 output
    v.act_elems = (size_t)B * (stem > l1 ? stem : l1);
    // the largest im2col matrix: stem3's input (H/2 x W/2, width/2 channels) or layer1's conv2 (H/4 x W/4, width)
    const size_t c_stem = (size_t)(cfg->img_h / 2) * (cfg->img_w / 2) * 9 * (size_t)(cfg->width / 2);
    const size_t c_l1 = (size_t)(cfg->img_h / 4) * (cfg->img_w / 4) * 9 * (size_t)cfg->width;
    v.col_elems = (size_t)B * (c_stem > c_l1 ? c_stem : c_l1);
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off += align_up(bytes, 256);
        return o;
    };
    for (int i = 0; i < 5; ++i) v.act[i] = take(v.act_elems * 4);
    v.col = take(v.col_elems * 4);
    v.mean = take((size_t)B * v.E * 4);
    v.tok = take((size_t)B * v.T * v.E * 4);
    v.tok0 = take((size_t)B * v.E * 4);
    v.q = take((size_t)B * v.E * 4);
    v.k = take((size_t)B * v.T * v.E * 4);
    v.v = take((size_t)B * v.T * v.E * 4);
    v.att = take((size_t)B * v.E * 4);
    v.proj = take((size_t)B * cfg->out_dim * 4);
    v.total = off;
    return v;
}

int check_cfg_f32(const mpreid_rn50_cfg *c) {
    ARG_CHECK(c != nullptr);
    ARG_CHECK(c->img_h > 0 && c->img_w > 0 && c->img_h % 32 == 0 && c->img_w % 32 == 0);
    ARG_CHECK(c->width >= 8 && c->width % 8 == 0 && c->n_blocks >= 4);
    ARG_CHECK(c->heads > 0 && (c->width * 32) % c->heads == 0 && (c->width * 32) / c->heads == 64);
    ARG_CHECK(c->out_dim > 0);
    return 0;
}

// one folded convolution (+ ReLU; or + identity + ReLU when res != 0: the identity is what `out` holds on entry)
int conv_f32(const mpreid_rn50_conv_f32 &c, const float *in, int B, int H, int W, int relu, int res, float *out, float *col,
             hipStream_t stream) {
    const int64_t M = (int64_t)B * H * W;
    const float *A = in;
    if (c.taps == 9) {
        const int64_t threads = M * 9 * (c.cin / 4);
        hipLaunchKernelGGL(im2col3x3_f32_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, in, B, H, W, c.cin,
                           col);
        LAUNCH_CHECK();
        A = col;
    }
    const int epi = res ? F32_LIN_RES_RELU : (relu ? F32_LIN_RELU : F32_LIN);
    return mpreid_gemm_f32_linear(A, c.w, M, c.cout, c.taps * c.cin, c.bias, out, c.cout, epi, stream);
}

int pool_f32(const float *in, int B, int H, int W, int C, float *out, hipStream_t stream) {
    const int64_t threads = (int64_t)B * (H / 2) * (W / 2) * C;
    hipLaunchKernelGGL(avgpool2_f32_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, in, B, H, W, C, out);
    LAUNCH_CHECK();
    return 0;
}

} // namespace

extern "C" size_t mpreid_rn50_workspace_bytes_f32(const mpreid_rn50_cfg *cfg, int batch) {
    if (check_cfg_f32(cfg) || batch <= 0) return 0;
    return layout_f32(cfg, batch).total;
}

extern "C" int mpreid_rn50_forward_f32(const mpreid_rn50_cfg *cfg, const mpreid_rn50_weights_f32 *w, const float *img, int B,
                                       float *out, void *ws, size_t ws_bytes, mpreid_stream_t stream_) {
    int rc = check_cfg_f32(cfg);
    if (rc) return rc;
    ARG_CHECK(w && img && out && B > 0 && w->blocks && w->stem1_w && w->stem1_b && w->q_w && w->k_w && w->v_w && w->c_w);
    const LayoutF32 v = layout_f32(cfg, B);
    if (!ws || ws_bytes < v.total) {
        mpreid_set_error("rn50 fp32 workspace too small: %zu < %zu", ws_bytes, v.total);
        return MPREID_ERR_WORKSPACE;
    }
    hipStream_t stream = (hipStream_t)stream_;
    char *base = (char *)ws;
    float *buf[5];
    for (int i = 0; i < 5; ++i) buf[i] = (float *)(base + v.act[i]);
    float *col = (float *)(base + v.col);

    // ---- stem (model/clip/model.py:128-134) ----
    int H = cfg->img_h / 2, W = cfg->img_w / 2;
    {
        const int c1 = cfg->width / 2;
        const int64_t threads = (int64_t)B * H * W * c1;
        hipLaunchKernelGGL(stem1_f32_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, img, w->stem1_w,
                           w->stem1_b, c1, B, cfg->img_h, cfg->img_w, buf[0]);
        LAUNCH_CHECK();
    }
    ARG_CHECK(w->stem2.taps == 9 && w->stem3.taps == 9 && w->stem2.cin % 4 == 0 && w->stem3.cin % 4 == 0);
    if ((rc = conv_f32(w->stem2, buf[0], B, H, W, 1, 0, buf[1], col, stream))) return rc;
    if ((rc = conv_f32(w->stem3, buf[1], B, H, W, 1, 0, buf[0], col, stream))) return rc;
    if ((rc = pool_f32(buf[0], B, H, W, w->stem3.cout, buf[1], stream))) return rc;
    H /= 2;
    W /= 2;
    int xi = 1;

    // ---- residual layers (model/clip/model.py:39-53) ----
    for (int bi = 0; bi < cfg->n_blocks; ++bi) {
        const mpreid_rn50_block_f32 &blk = w->blocks[bi];
        ARG_CHECK(blk.stride == 1 || blk.stride == 2);
        ARG_CHECK(blk.conv1.taps == 1 && blk.conv2.taps == 9 && blk.conv3.taps == 1 && blk.conv2.cin % 4 == 0);
        int free_i[4], nf = 0;
        for (int i = 0; i < 5; ++i)
            if (i != xi) free_i[nf++] = i;
        float *x = buf[xi], *t1 = buf[free_i[0]], *t2 = buf[free_i[1]], *t3 = buf[free_i[2]], *t4 = buf[free_i[3]];
        if ((rc = conv_f32(blk.conv1, x, B, H, W, 1, 0, t1, col, stream))) return rc;
        if ((rc = conv_f32(blk.conv2, t1, B, H, W, 1, 0, t2, col, stream))) return rc;
        int OH = H, OW = W;
        const float *o2 = t2;
        if (blk.stride == 2) {
            if ((rc = pool_f32(t2, B, H, W, blk.conv2.cout, t1, stream))) return rc;
            o2 = t1;
            OH = H / 2;
            OW = W / 2;
        }
        // the identity goes INTO the destination buffer, the conv3 GEMM then computes relu(dst + conv3 + bias) in place
        float *dst;
        if (blk.down.w) {
            const float *xin = x;
            if (blk.stride == 2) {
                if ((rc = pool_f32(x, B, H, W, blk.down.cin, t3, stream))) return rc;
                xin = t3;
            }
            if ((rc = conv_f32(blk.down, xin, B, OH, OW, 0, 0, t4, col, stream))) return rc;
            dst = t4;
            xi = free_i[3];
        } else {
            ARG_CHECK(blk.stride == 1 && blk.conv3.cout == blk.conv1.cin);
            dst = x;      // x itself is the identity and is not needed afterwards
        }
        if ((rc = conv_f32(blk.conv3, o2, B, OH, OW, 1, 1, dst, col, stream))) return rc;
        H = OH;
        W = OW;
    }
    ARG_CHECK(H * W == v.S && w->blocks[cfg->n_blocks - 1].conv3.cout == v.E);

    // ---- attention pool (model/clip/model.py:56-90; only the output at token 0 is used) + head ----
    float *mean = (float *)(base + v.mean), *tok = (float *)(base + v.tok), *tok0 = (float *)(base + v.tok0);
    float *q = (float *)(base + v.q), *k = (float *)(base + v.k), *vv = (float *)(base + v.v);
    float *att = (float *)(base + v.att), *proj = (float *)(base + v.proj);
    hipLaunchKernelGGL(tokens_f32_kernel, dim3(B), dim3(256), 0, stream, buf[xi], w->pos_emb, v.S, v.E, mean, tok);
    hipLaunchKernelGGL(gather_tok0_f32_kernel, dim3(B), dim3(256), 0, stream, tok, v.T, v.E, tok0);
    LAUNCH_CHECK();
    if ((rc = mpreid_gemm_f32_linear(tok0, w->q_w, B, v.E, v.E, w->q_b, q, v.E, F32_LIN, stream))) return rc;
    if ((rc = mpreid_gemm_f32_linear(tok, w->k_w, (int64_t)B * v.T, v.E, v.E, w->k_b, k, v.E, F32_LIN, stream))) return rc;
    if ((rc = mpreid_gemm_f32_linear(tok, w->v_w, (int64_t)B * v.T, v.E, v.E, w->v_b, vv, v.E, F32_LIN, stream))) return rc;
    hipLaunchKernelGGL(pool_attend_f32_kernel, dim3((unsigned)(B * cfg->heads)), dim3(64), (size_t)v.T * 4, stream, q, k, vv, v.T,
                       v.E, cfg->heads, att);
    LAUNCH_CHECK();
    if ((rc = mpreid_gemm_f32_linear(att, w->c_w, B, cfg->out_dim, v.E, w->c_b, proj, cfg->out_dim, F32_LIN, stream))) return rc;
    hipLaunchKernelGGL(head_f32_kernel, dim3(B), dim3(256), 0, stream, mean, proj, v.E, cfg->out_dim, w->bn_scale, w->bn_shift, out);
    LAUNCH_CHECK();
    return 0;
}
