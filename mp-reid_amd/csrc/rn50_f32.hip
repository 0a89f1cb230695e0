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
#include "conv_f16.h"   // ... and its 3x3 convolutions as implicit GEMMs over fp16 pairs
#include "gemm_f16.h"   // the split-precision tower (below) runs its layer1-4 convolutions on the fp16 matrix cores

int mpreid_gemm_f32_linear(const float *A, const float *Wt, int64_t M, int64_t N, int K, const float *bias, float *C,
                           int64_t ldc, int epi, hipStream_t stream);
enum { F32_LIN = 2, F32_LIN_RES = 4, F32_LIN_RELU = 5, F32_LIN_RES_RELU = 6 };   // distance.hip: EPI_LIN*

namespace {

// The stem's input: fp32 [B][3][H][W] (val_transforms applied), or -- img8 != NULL -- uint8 [B][H][W][3] (what the loader has
// after Resize) with ToTensor + Normalize of datasets/make_dataloader.py:57-61 applied on the fly, in the reference's order and
// rounding: (x / 255 - mean) / std with two correctly rounded divisions (same bits as the host-transformed fp32 tensor).
// view: the test-time-augmentation views of processor/processor_uniprompt_stage2.py:605-633 applied to the NORMALISED image while
// it is read (0 original, 1 torch.flip(img, [3]), 2 pseudo-IR img.mean(dim=1) in all channels = ((c0 + c1) + c2) / 3, 3 pseudo-RGB
// channel 0 in all channels) -- what include/mpreid.h calls MPREID_VIEW_*; same bits as the view tensor materialised on the
// HOST (torch CPU: sum, then a true division -- the reference's CPU path); torch's DEVICE mean multiplies by a rounded 1/3 and
// is up to one ulp apart per pseudo-IR pixel (include/mpreid.h, MPREID_VIEW_PSEUDO_IR).
struct StemIn {
    const float *img;
    const uint8_t *img8;
    float mean[3], sd[3];
    int view;
};
__device__ __forceinline__ float stem_px_raw(const StemIn &in, int b, int c, int iy, int ix, int H, int W) {
    if (in.img8) return __fdiv_rn(__fdiv_rn((float)in.img8[(((int64_t)b * H + iy) * W + ix) * 3 + c], 255.0f) - in.mean[c], in.sd[c]);
    return in.img[(((int64_t)b * 3 + c) * H + iy) * W + ix];
}
__device__ __forceinline__ float stem_px(const StemIn &in, int b, int c, int iy, int ix, int H, int W) {
    if (in.view == 0) return stem_px_raw(in, b, c, iy, ix, H, W);
    const int x = in.view == 1 ? W - 1 - ix : ix;
    if (in.view == 2)
        return __fdiv_rn((stem_px_raw(in, b, 0, iy, x, H, W) + stem_px_raw(in, b, 1, iy, x, H, W)) + stem_px_raw(in, b, 2, iy, x, H, W), 3.0f);
    return stem_px_raw(in, b, in.view == 3 ? 0 : c, iy, x, H, W);
}

// stem conv1 + bn1 + relu: [B][3][H][W] fp32 -> [B][H/2][W/2][cout] fp32 (stride 2, pad 1); w [cout][c][kh][kw] folded
__global__ __launch_bounds__(256) void stem1_f32_kernel(const StemIn img, const float *__restrict__ w,
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
                const float x = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? stem_px(img, b, c, iy, ix, H, W) : 0.0f;
                acc = fmaf(x, w[n * 27 + c * 9 + kh * 3 + kw], acc);
            }
    const float v = acc + bias[n];
    out[gid] = v < 0.0f ? 0.0f : v;
}

// the same convolution, one thread per OUTPUT PIXEL (all COUT channels): the 27 inputs are read once instead of once per
// output channel, the folded weights sit in LDS ([27][COUT], broadcast reads), the pixel's COUT results leave as one
// contiguous run.  Same arithmetic per output (fmaf chain over (c, kh, kw) ascending, + bias, ReLU): same bits as
// stem1_f32_kernel, which stays for channel counts other than 32 / 16 / 8.
template <int COUT>
__global__ __launch_bounds__(256) void stem1_px_kernel(const StemIn img, const float *__restrict__ w,
                                                       const float *__restrict__ bias, int B, int H, int W,
                                                       float *__restrict__ out) {
    __shared__ float ws[27 * COUT + COUT];
    for (int t = threadIdx.x; t < 27 * COUT; t += 256) ws[(t % 27) * COUT + t / 27] = w[t];   // [n][27] -> [27][n]
    for (int t = threadIdx.x; t < COUT; t += 256) ws[27 * COUT + t] = bias[t];
    __syncthreads();
    const int OH = H / 2, OW = W / 2;
    const int64_t npix = (int64_t)B * OH * OW;
    const int64_t pix_raw = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t pix = pix_raw < npix ? pix_raw : npix - 1;   // (the lanes past the end compute the last pixel again and store nothing)
    const int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH), b = (int)(pix / ((int64_t)OW * OH));
    float acc[COUT];
#pragma unroll
    for (int n = 0; n < COUT; ++n) acc[n] = 0.0f;
    // the 27 inputs first (27 loads in flight), then tap by tap: COUT weights from LDS (broadcast reads), COUT FMAs.  The
    // scheduling barrier per tap keeps the compiler from hoisting all 27 x COUT weight reads to the top: it did, and the
    // kernel ran with 384 spilled registers at one wave per SIMD (1.5 ms at B = 256).
    float xin[27];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int iy = oy * 2 + kh - 1, ix = ox * 2 + kw - 1;
                xin[c * 9 + kh * 3 + kw] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? stem_px(img, b, c, iy, ix, H, W) : 0.0f;
            }
#pragma unroll
    for (int t = 0; t < 27; ++t) {   // t = c * 9 + kh * 3 + kw: the accumulation order of the reference chain
        const float *wr = ws + t * COUT;
#pragma unroll
        for (int n = 0; n < COUT; ++n) acc[n] = fmaf(xin[t], wr[n], acc[n]);
        __builtin_amdgcn_sched_barrier(0);
    }
    // The pixel's COUT results go through a wave-private LDS patch so that a store instruction covers 1 KB of consecutive
    // bytes (lane = 16-byte piece of the wave's 64 x COUT block): written straight from the registers every instruction
    // touched 64 different lines, 16 bytes each, and the kernel ran at 180 GB/s of output (1.46 ms at B = 256).
    constexpr int PS = COUT + 4;   // patch row stride in floats (16-byte aligned rows, spread over the banks)
    __shared__ __attribute__((aligned(16))) float patch[4][64 * PS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *pw = patch[wave];
#pragma unroll
    for (int n = 0; n < COUT; n += 4) {
        float4 v;
        v.x = acc[n] + ws[27 * COUT + n];
        v.y = acc[n + 1] + ws[27 * COUT + n + 1];
        v.z = acc[n + 2] + ws[27 * COUT + n + 2];
        v.w = acc[n + 3] + ws[27 * COUT + n + 3];
        v.x = v.x < 0.0f ? 0.0f : v.x;
        v.y = v.y < 0.0f ? 0.0f : v.y;
        v.z = v.z < 0.0f ? 0.0f : v.z;
        v.w = v.w < 0.0f ? 0.0f : v.w;
        *reinterpret_cast<float4 *>(pw + lane * PS + n) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    constexpr int CPP = COUT / 4;                               // 16-byte pieces per pixel
    const int64_t pix0 = (int64_t)blockIdx.x * 256 + wave * 64;   // first pixel of the wave
#pragma unroll
    for (int it = 0; it < CPP; ++it) {
        const int idx = it * 64 + lane;
        const int p = idx / CPP, ch = idx - p * CPP;
        if (pix0 + p < npix)
            *reinterpret_cast<float4 *>(out + (pix0 + p) * COUT + ch * 4) = *reinterpret_cast<const float4 *>(pw + p * PS + ch * 4);
    }
}

static int launch_stem1(const StemIn &img, const float *w, const float *bias, int c1, int B, int H, int W, float *out,
                        hipStream_t stream) {
    const int64_t px = (int64_t)B * (H / 2) * (W / 2);
    const dim3 gp((unsigned)((px + 255) / 256));
    if (c1 == 32) hipLaunchKernelGGL(stem1_px_kernel<32>, gp, dim3(256), 0, stream, img, w, bias, B, H, W, out);
    else if (c1 == 16) hipLaunchKernelGGL(stem1_px_kernel<16>, gp, dim3(256), 0, stream, img, w, bias, B, H, W, out);
    else if (c1 == 8) hipLaunchKernelGGL(stem1_px_kernel<8>, gp, dim3(256), 0, stream, img, w, bias, B, H, W, out);
    else {
        const int64_t threads = px * c1;
        hipLaunchKernelGGL(stem1_f32_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, img, w, bias, c1, B, H, W,
                           out);
    }
    LAUNCH_CHECK();
    return 0;
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
template <bool RELU>   // RELU: x4 holds pre-activations (the split tower never writes a block's closing ReLU back)
__global__ __launch_bounds__(256) void tokens_f32_kernel(const float *__restrict__ x4, const float *__restrict__ pos, int S, int E,
                                                         float *__restrict__ mean, float *__restrict__ tok) {
    // grid (B, ceil(E / 256)): one thread per (image, channel) -- with one workgroup per image (the first version) a
    // 256-image batch ran on 256 workgroups of 4 waves, each thread walking 8 channels x S tokens one after the other: 0.46 ms
    // for 0.54 GB.  Same sum order per channel: same bits.
    const int b = blockIdx.x;
    for (int c = blockIdx.y * 256 + threadIdx.x; c < E; c += 256 * gridDim.y) {
        float sum = 0.0f;
        for (int t = 0; t < S; ++t) {
            float v = x4[((int64_t)b * S + t) * E + c];
            if (RELU) v = v < 0.0f ? 0.0f : v;
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

static StemIn stem_in(const float *img, const uint8_t *img8, const float *mean, const float *sd, int view = 0) {
    StemIn s{img, img8, {0.f, 0.f, 0.f}, {1.f, 1.f, 1.f}, view};
    if (img8)
        for (int c = 0; c < 3; ++c) {
            s.mean[c] = mean[c];
            s.sd[c] = sd[c];
        }
    return s;
}

static int rn50_forward_f32_impl(const mpreid_rn50_cfg *cfg, const mpreid_rn50_weights_f32 *w, const StemIn &img, int B,
                                 float *out, void *ws, size_t ws_bytes, mpreid_stream_t stream_) {
    int rc = check_cfg_f32(cfg);
    if (rc) return rc;
    ARG_CHECK(w && (img.img || img.img8) && out && B > 0 && w->blocks && w->stem1_w && w->stem1_b && w->q_w && w->k_w && w->v_w && w->c_w);
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
    if ((rc = launch_stem1(img, w->stem1_w, w->stem1_b, cfg->width / 2, B, cfg->img_h, cfg->img_w, buf[0], stream))) return rc;
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
    hipLaunchKernelGGL(tokens_f32_kernel<false>, dim3(B, (v.E + 255) / 256), dim3(256), 0, stream, buf[xi], w->pos_emb, v.S, v.E, mean, tok);
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

extern "C" int mpreid_rn50_forward_f32(const mpreid_rn50_cfg *cfg, const mpreid_rn50_weights_f32 *w, const float *img, int B,
                                       float *out, void *ws, size_t ws_bytes, mpreid_stream_t stream_) {
    ARG_CHECK(img);
    return rn50_forward_f32_impl(cfg, w, stem_in(img, nullptr, nullptr, nullptr), B, out, ws, ws_bytes, stream_);
}

extern "C" int mpreid_rn50_forward_f32_u8(const mpreid_rn50_cfg *cfg, const mpreid_rn50_weights_f32 *w, const uint8_t *img_hwc,
                                          const float *mean, const float *stdv, int B, float *out, void *ws, size_t ws_bytes,
                                          mpreid_stream_t stream_) {
    ARG_CHECK(img_hwc && mean && stdv);
    return rn50_forward_f32_impl(cfg, w, stem_in(nullptr, img_hwc, mean, stdv), B, out, ws, ws_bytes, stream_);
}

extern "C" int mpreid_rn50_forward_f32_view(const mpreid_rn50_cfg *cfg, const mpreid_rn50_weights_f32 *w, const float *img_f32,
                                            const uint8_t *img_hwc, const float *mean, const float *stdv, int view, int B,
                                            float *out, void *ws, size_t ws_bytes, mpreid_stream_t stream_) {
    ARG_CHECK((img_f32 != nullptr) != (img_hwc != nullptr) && (!img_hwc || (mean && stdv)) && view >= 0 && view <= 3);
    return rn50_forward_f32_impl(cfg, w, stem_in(img_f32, img_hwc, mean, stdv, view), B, out, ws, ws_bytes, stream_);
}


// =============================================================================================================
// SPLIT-precision mode of the RN50 tower (MODEL.NAME 'RN50' + MODEL.ENCODER_PRECISION 'split', round 4): the ViT's recipe
// carried over.  Activations stay fp32 NHWC in HBM (the "residual stream" of a ResNet is every tensor); every convolution
// of layer1-4 and the attention pool's k / v projections -- 95 % of the tower's FLOPs -- run as GEMMs over fp16 PAIRS
// x = hi + lo on the fp16 matrix cores, three products hi.hi' + lo.hi' + hi.lo' per multiply-add with fp32 accumulation
// (gemm_f16.hip GE_S_BIAS_F32 / GE_S_BIAS_RES: the kernels the ViT's split mode uses, unchanged):
//   1x1 convolution   pack_pairs_kernel: fp32 [M][C] -> pairs [Mp][hi(kseg) | lo(kseg)] (ReLU of the producer applied on the
//                     way), then out fp32 = acc * 2^-e + bias'                      (BatchNorm folded into W 2^e and bias');
//                     conv1 writes relu(.) directly as conv2's pair operand instead (GE_S_BIAS_RELU_PAIR)
//   3x3 convolution   the same pack (channels padded to 64), then an IMPLICIT GEMM over the nine shifted views of the pair
//                     tensor (conv_f16.hip, pair form: per (tap, 64 channels) the three products; padding pixels read a page
//                     of zeros) -- the materialised im2col pair matrix of the first version cost 9x the activation bytes,
//                     written and read: 21 % of the forward
//   conv3 + identity  GE_S_BIAS_RES: the identity (block input, or the downsample branch's output) sits in the destination,
//                     dst += acc * 2^-e + bias'; the block's closing ReLU is never written: every reader of dst applies it
//                     (the next block's packs and average pools on read, its conv3 through GemmArgs::relu_x when dst is
//                     its identity); where no average pool sits between conv2 and conv3, conv2's epilogue writes conv3's
//                     ReLU-ed pair operand directly, and conv3 writes relu(dst) as the next block's conv1 operand beside the fp32
//                     identity (GE_S_BIAS_RES_PAIR)
// The stem's first convolution (K = 27: direct fp32 FMAs), the one-query attention and the 1-row projections q / c stay on
// the exact fp32 path above.  Channel counts are padded to the GEMM's granularity (kseg to 64, cout to 128: the stem's and
// layer1's 32 / 64-channel tensors are stored with a row stride of 128).
// =============================================================================================================
namespace {

// fp32 [rows][ld_in] (first C channels real) -> fp16 pairs [rows_pad][2 * kseg]: hi at k, lo at kseg + k, zeros for k >= C
// and for rows >= rows.  RELU: max(x, 0) first (the tensor itself keeps its pre-activation values: every reader applies the
// ReLU on the way).  One thread per 4 channels of kseg.
template <bool RELU>
__global__ __launch_bounds__(256) void pack_pairs_kernel(const float *__restrict__ in, int64_t rows, int64_t rows_pad, int C, int ld_in,
                                                         int kseg, _Float16 *__restrict__ out) {
    const int k4n = kseg / 4;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= rows_pad * k4n) return;
    const int k4 = (int)(gid % k4n);
    const int64_t r = gid / k4n;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (r < rows && k4 * 4 < C) {   // C % 4 == 0
        const float4 x = *reinterpret_cast<const float4 *>(in + r * ld_in + k4 * 4);
        v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
        if (RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] < 0.f ? 0.f : v[e];
        }
    }
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    h4 hi, lo;
    split_pairs<4>(v, hi, lo);
    _Float16 *o = out + r * 2 * kseg + k4 * 4;
    *reinterpret_cast<h4 *>(o) = hi;
    *reinterpret_cast<h4 *>(o + kseg) = lo;
}

// AvgPool2d(2) on NHWC fp32 with channel strides (ld_in -> ld_out), optional ReLU on read
template <bool RELU>
__global__ __launch_bounds__(256) void avgpool2_ld_kernel(const float *__restrict__ in, int B, int H, int W, int C, int ld_in,
                                                          int ld_out, float *__restrict__ out) {
    const int OH = H / 2, OW = W / 2;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (int64_t)B * OH * OW * C) return;
    const int c = (int)(gid % C);
    const int64_t pix = gid / C;
    const int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH), b = (int)(pix / ((int64_t)OW * OH));
    const float *p = in + (((int64_t)b * H + oy * 2) * W + ox * 2) * ld_in + c;
    float a0 = p[0], a1 = p[ld_in], a2 = p[(int64_t)W * ld_in], a3 = p[(int64_t)W * ld_in + ld_in];
    if (RELU) {
        a0 = a0 < 0.f ? 0.f : a0;
        a1 = a1 < 0.f ? 0.f : a1;
        a2 = a2 < 0.f ? 0.f : a2;
        a3 = a3 < 0.f ? 0.f : a3;
    }
    out[pix * ld_out + c] = (((a0 + a1) + a2) + a3) * 0.25f;
}


struct LayoutSplit {
    LayoutF32 f;           // the fp32 part: five activation buffers (sized for padded channel strides), stem im2col, pool buffers
    size_t pairs, pairs2, total;   // pair matrices: the GEMM A operand of the layer in flight; the operand a 3x3 convolution
                                   // writes for the 1x1 behind it
    size_t act_elems;
};

inline int64_t pad_rows(int64_t m) { return (int64_t)align_up((size_t)m, 256); }

LayoutSplit layout_split(const mpreid_rn50_cfg *cfg, int B) {
    LayoutSplit v{};
    const int fh = cfg->img_h / 16, fw = cfg->img_w / 16;
    const int S = fh * fw, T = S + 1, E = cfg->width * 32;
    const int wd = cfg->width;
    auto np = [](int c) { return (int64_t)align_up((size_t)c, 128); };
    // activation buffers: the widest tensors are the stem's (H/2 x W/2 x width), layer1's (H/4 x W/4 x 4 width, strides padded to
    // 128) and -- per pixel count -- nothing later is larger
    const int64_t px2 = (int64_t)(cfg->img_h / 2) * (cfg->img_w / 2), px4 = px2 / 4;
    // GEMM outputs have their rows padded to 256 and their channel stride to 128.  Widest tensor per resolution: the stem's
    // (H/2 x W/2 x width); H/4: layer1's output 4 width; H/8: layer2's 8 width; H/16: layer4's 32 width
    int64_t a = pad_rows((int64_t)B * px2) * np(wd);
    a = std::max<int64_t>(a, pad_rows((int64_t)B * px4) * np(wd * 4));
    a = std::max<int64_t>(a, pad_rows((int64_t)B * px4 / 4) * np(wd * 8));
    a = std::max<int64_t>(a, pad_rows((int64_t)B * S) * np(wd * 32));
    v.act_elems = (size_t)a;
    // pair matrix: the largest A operand.  A block's conv1 / conv2 run at the block's INPUT resolution (the stride is an
    // average pool behind conv2): H/4: 1x1 over <= 4 width, 3x3 over <= 2 width; H/8: 8 width / 4 width; H/16: 32 width / 8 width
    // (a 3x3 convolution's A operand is the plain pair tensor too: the nine taps are shifted views of it)
    auto kseg = [](int64_t k) { return (int64_t)align_up((size_t)k, 64); };
    int64_t pe = pad_rows((int64_t)B * px4) * 2 * std::max(kseg(wd * 2), kseg(wd * 4));
    pe = std::max<int64_t>(pe, pad_rows((int64_t)B * px2) * 2 * kseg(wd));             // the stem's conv2 / conv3
    pe = std::max<int64_t>(pe, pad_rows((int64_t)B * px4 / 4) * 2 * std::max(kseg(wd * 4), kseg(wd * 8)));
    pe = std::max<int64_t>(pe, pad_rows((int64_t)B * S) * 2 * std::max(kseg(wd * 8), kseg(wd * 32)));
    pe = std::max<int64_t>(pe, pad_rows((int64_t)B * T) * 2 * kseg(E));
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off += align_up(bytes, 256);
        return o;
    };
    for (int i = 0; i < 5; ++i) v.f.act[i] = take(v.act_elems * 4);
    v.f.col = take(256);   // (the fp32 path's im2col buffer is not used by this mode)
    v.f.S = S; v.f.T = T; v.f.E = E;
    v.f.mean = take((size_t)B * E * 4);
    v.f.tok = take((size_t)pad_rows((int64_t)B * T) * E * 4);
    v.f.tok0 = take((size_t)B * E * 4);
    v.f.q = take((size_t)B * E * 4);
    v.f.k = take((size_t)pad_rows((int64_t)B * T) * E * 4);
    v.f.v = take((size_t)pad_rows((int64_t)B * T) * E * 4);
    v.f.att = take((size_t)B * E * 4);
    v.f.proj = take((size_t)B * cfg->out_dim * 4);
    v.pairs = take((size_t)pe * 2);
    v.pairs2 = take((size_t)pe * 2);
    v.total = off;
    return v;
}

struct ActView { float *p; int C, ld; bool dirty; };   // fp32 NHWC tensor: C real channels, row stride ld, ReLU pending?

// one folded convolution of the split tower: pairs of `in` (ReLU applied on read when in.dirty) -> GEMM; res != 0: out += (GE_S_BIAS_RES)
// pair_out (3x3 only): the result leaves as the ReLU-ed pair operand of the NEXT pair GEMM ([Mp][2 * pair_c]) instead of fp32;
// prepacked (1x1 only): `in` has been delivered that way already -- no pack pass
int conv_split(const mpreid_rn50_conv_split &c, ActView &in, int B, int H, int W, int res, float *out, _Float16 *pairs,
               hipStream_t stream, const _Float16 *zero_page = nullptr, _Float16 *pair_out = nullptr,
               int pair_c = 0, const _Float16 *prepacked = nullptr, const _Float16 *act_pairs = nullptr) {
    const int64_t M = (int64_t)B * H * W, Mp = pad_rows(M);
    // kseg: 1x1 -- the padded K of the pair matrix; 3x3 -- the padded channel count of ONE tap (the weights are laid out
    // for the implicit GEMM, include/mpreid.h)
    ARG_CHECK(c.w && c.bias && c.cin == in.C && c.cin % 4 == 0 && c.kseg % 64 == 0 && c.kseg >= c.cin &&
              (c.taps == 9 || c.kseg >= c.taps * c.cin) && c.npad % 128 == 0 && c.npad >= c.cout && (c.taps == 1 || c.taps == 9));
    const int64_t threads = Mp * (c.kseg / 4);
    const dim3 grid((unsigned)((threads + 255) / 256));
    if (c.taps == 9) {
        // 3x3: the input as pairs [pixel][hi(C) | lo(C)] (ReLU applied on the way), then the implicit GEMM over the nine shifted
        // views of it (conv_f16.hip, pair form) -- no im2col matrix (it cost 9x the activation bytes, written and read)
        ARG_CHECK(res == 0 && zero_page != nullptr);
        if (act_pairs)   // the producer wrote this convolution's pair operand itself (GE_S_BIAS_RELU_PAIR)
            pairs = const_cast<_Float16 *>(act_pairs);
        else if (in.dirty)
            hipLaunchKernelGGL((pack_pairs_kernel<true>), grid, dim3(256), 0, stream, in.p, M, Mp, in.C, in.ld, c.kseg, pairs);
        else
            hipLaunchKernelGGL((pack_pairs_kernel<false>), grid, dim3(256), 0, stream, in.p, M, Mp, in.C, in.ld, c.kseg, pairs);
        LAUNCH_CHECK();
        ConvArgs a{};
        a.act = pairs;
        a.wgt = (const _Float16 *)c.w;
        a.bias = c.bias;
        a.zero_page = zero_page;
        a.H = H; a.W = W; a.C = c.kseg;
        a.M = (int)M;
        a.N = c.cout; a.Npad = c.npad;
        a.ldo = c.npad;
        a.taps = 9;
        a.split = 1;
        a.oscale = c.oscale;
        a.out32 = out;
        a.out_pairs = pair_out;
        a.pair_c = pair_c;
        return launch_conv_f16(a, stream);
    } else if (prepacked) {
        pairs = const_cast<_Float16 *>(prepacked);
    } else if (in.dirty) {
        hipLaunchKernelGGL((pack_pairs_kernel<true>), grid, dim3(256), 0, stream, in.p, M, Mp, in.C, in.ld, c.kseg, pairs);
    } else {
        hipLaunchKernelGGL((pack_pairs_kernel<false>), grid, dim3(256), 0, stream, in.p, M, Mp, in.C, in.ld, c.kseg, pairs);
    }
    LAUNCH_CHECK();
    GemmArgs g{};
    g.A = pairs;
    g.W = (const _Float16 *)c.w;
    g.M = (int)Mp;
    g.N = c.npad;
    g.K = 2 * c.kseg;
    g.kseg = c.kseg;
    g.oscale = c.oscale;
    g.out = out;
    g.ldo = c.npad;
    g.bias = c.bias;
    g.relu_x = res == 2;   // the destination is a block input whose ReLU is still pending (it was never written back)
    if (pair_out && res) {   // block output: the fp32 identity AND relu(.) as the pair operand of the next block's first convolution
        ARG_CHECK(pair_c == c.npad);
        g.pair_out = pair_out;
        g.pair_c = pair_c;
        return launch_gemm_f16(g, GE_S_BIAS_RES_PAIR, stream);
    }
    if (pair_out) {        // 1x1 whose only consumer is a pair convolution: relu(.) as that convolution's operand, no fp32 tensor
        ARG_CHECK(res == 0 && pair_c % 64 == 0 && pair_c >= c.cout);
        g.out = pair_out;
        g.ldo = 2 * (int64_t)pair_c;
        g.pair_c = pair_c;
        return launch_gemm_f16(g, GE_S_BIAS_RELU_PAIR, stream);
    }
    return launch_gemm_f16(g, res ? GE_S_BIAS_RES : GE_S_BIAS_F32, stream);
}

} // namespace

extern "C" size_t mpreid_rn50_workspace_bytes_split(const mpreid_rn50_cfg *cfg, int batch) {
    if (check_cfg_f32(cfg) || batch <= 0) return 0;
    return layout_split(cfg, batch).total;
}

static int rn50_forward_split_impl(const mpreid_rn50_cfg *cfg, const mpreid_rn50_weights_split *w, const StemIn &img, int B,
                                   float *out, void *ws, size_t ws_bytes, mpreid_stream_t stream_) {
    int rc = check_cfg_f32(cfg);
    if (rc) return rc;
    ARG_CHECK(w && (img.img || img.img8) && out && B > 0 && w->blocks && w->f32.stem1_w && w->f32.stem1_b && w->f32.q_w && w->f32.c_w && w->k.w && w->v.w &&
              w->stem2.w && w->stem3.w);
    const LayoutSplit v = layout_split(cfg, B);
    if (!ws || ws_bytes < v.total) {
        mpreid_set_error("rn50 split workspace too small: %zu < %zu", ws_bytes, v.total);
        return MPREID_ERR_WORKSPACE;
    }
    hipStream_t stream = (hipStream_t)stream_;
    char *base = (char *)ws;
    float *buf[5];
    for (int i = 0; i < 5; ++i) buf[i] = (float *)(base + v.f.act[i]);
    const _Float16 *zero_page = (const _Float16 *)(base + v.f.col);   // 256 bytes of zeros: the padding pixels of the 3x3 convolutions
    HIP_TRY(hipMemsetAsync(base + v.f.col, 0, 256, (hipStream_t)stream_));
    _Float16 *pairs = (_Float16 *)(base + v.pairs), *pairs2 = (_Float16 *)(base + v.pairs2);
    const mpreid_rn50_weights_f32 &wf = w->f32;

    // ---- stem on the exact fp32 path (ReLU applied by its GEMM epilogues) ----
    int H = cfg->img_h / 2, W = cfg->img_w / 2;
    if ((rc = launch_stem1(img, wf.stem1_w, wf.stem1_b, cfg->width / 2, B, cfg->img_h, cfg->img_w, buf[0], stream))) return rc;
    // stem conv2 / conv3 (3x3 over width/2 channels): pair GEMMs like the layers' (their outputs are stored with a channel
    // stride of 128; on the exact fp32 path these two narrow layers -- N = 32 and 64 in 128-wide tiles -- cost 4.7 ms of a
    // 25 ms forward at B = 256)
    ARG_CHECK(w->stem2.taps == 9 && w->stem3.taps == 9);
    {
        ActView s1{buf[0], cfg->width / 2, cfg->width / 2, true};   // (stem1 applied its ReLU already: max(x, 0) again is the identity)
        if ((rc = conv_split(w->stem2, s1, B, H, W, 0, buf[1], pairs, stream, zero_page))) return rc;
        ActView s2{buf[1], w->stem2.cout, w->stem2.npad, true};
        if ((rc = conv_split(w->stem3, s2, B, H, W, 0, buf[2], pairs, stream, zero_page))) return rc;
        const int64_t threads = (int64_t)B * (H / 2) * (W / 2) * w->stem3.cout;
        hipLaunchKernelGGL((avgpool2_ld_kernel<true>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, buf[2], B, H, W,
                           w->stem3.cout, w->stem3.npad, w->stem3.cout, buf[1]);
        LAUNCH_CHECK();
    }
    H /= 2;
    W /= 2;
    int xi = 1;
    ActView x{buf[1], w->stem3.cout, w->stem3.cout, false};
    _Float16 *bufA = pairs, *bufB = pairs2;   // pair buffers: bufA = the current block's input pairs (see the loop)
    bool have_xp = false;                     // bufA already holds relu(x) as pairs (written by the previous block's conv3)

    // ---- residual layers on the fp16 matrix cores (pairs) ----
    for (int bi = 0; bi < cfg->n_blocks; ++bi) {
        const mpreid_rn50_block_split &blk = w->blocks[bi];
        ARG_CHECK(blk.stride == 1 || blk.stride == 2);
        ARG_CHECK(blk.conv1.taps == 1 && blk.conv2.taps == 9 && blk.conv3.taps == 1);
        int free_i[4], nf = 0;
        for (int i = 0; i < 5; ++i)
            if (i != xi) free_i[nf++] = i;
        float *t1 = buf[free_i[0]], *t2 = buf[free_i[1]], *t3 = buf[free_i[2]], *t4 = buf[free_i[3]];
        // conv1 reads x with the ReLU of the previous block's sum applied on the way.  The ReLU is NOT written back (that made the
        // pack a three-pass kernel): whoever reads x as the identity below applies it again -- the downsample branch's average
        // pool on read, conv3's residual epilogue through GemmArgs::relu_x.
        // Pair operands written by their producers (rows of a pair matrix are padded to 256: only when the pixel count is such
        // a multiple): conv1 -> conv2 (GE_S_BIAS_RELU_PAIR: no fp32 tensor, no pack pass), conv2 -> conv3 when no average pool
        // sits between them (13 of the 16 blocks), conv3 -> the next block's conv1 (GE_S_BIAS_RES_PAIR: the block output leaves
        // twice, as the fp32 identity and as relu(.) pairs).  Two pair buffers, bufA (this block's input pairs) and bufB: a
        // convolution never writes the one it reads, and the roles swap from block to block.
        const bool rows_ok = ((int64_t)B * H * W) % 256 == 0;
        const bool fuse12 = rows_ok && blk.conv2.kseg >= blk.conv1.cout && blk.conv2.cin == blk.conv1.cout;
        const bool fuse23 = rows_ok && blk.stride == 1 && blk.conv3.kseg >= blk.conv2.cout && blk.conv3.cin == blk.conv2.cout;
        if ((rc = conv_split(blk.conv1, x, B, H, W, 0, t1, bufA, stream, nullptr, fuse12 ? bufB : nullptr, blk.conv2.kseg,
                             have_xp ? bufA : nullptr)))
            return rc;   // (bufA now holds relu(x) as pairs [Mp][2 * conv1.kseg], packed here or written by the previous block)
        ActView a1{t1, blk.conv1.cout, blk.conv1.npad, true};
        int OH = H, OW = W;
        if (blk.stride == 2) {
            OH = H / 2;
            OW = W / 2;
        }
        // the downsample branch (it only needs x) runs before conv2: bufA is free again afterwards
        float *dst;
        if (blk.down.w) {
            if (blk.stride == 2) {
                const int64_t threads = (int64_t)B * OH * OW * x.C;
                if (x.dirty)
                    hipLaunchKernelGGL((avgpool2_ld_kernel<true>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, x.p, B, H,
                                       W, x.C, x.ld, x.C, t3);
                else
                    hipLaunchKernelGGL((avgpool2_ld_kernel<false>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, x.p, B, H,
                                       W, x.C, x.ld, x.C, t3);
                LAUNCH_CHECK();
                ActView xin{t3, x.C, x.C, false};
                if ((rc = conv_split(blk.down, xin, B, OH, OW, 0, t4, bufA, stream))) return rc;
            } else {
                // stride 1: the operand is relu(x) as pairs with conv1's k segment -- exactly what bufA holds
                ARG_CHECK(blk.down.kseg == blk.conv1.kseg && blk.down.cin == blk.conv1.cin);
                ActView xin = x;
                if ((rc = conv_split(blk.down, xin, B, OH, OW, 0, t4, bufA, stream, nullptr, nullptr, 0, bufA))) return rc;
            }
            dst = t4;
            xi = free_i[3];
        } else {
            ARG_CHECK(blk.stride == 1 && blk.conv3.cout == x.C && blk.conv3.npad == x.ld);
            dst = x.p;      // x itself is the identity and is not needed afterwards
        }
        _Float16 *c3_pairs = fuse12 ? bufA : bufB;   // conv2 reads bufB (fuse12) or packs into bufA: it writes the other one
        if ((rc = conv_split(blk.conv2, a1, B, H, W, 0, t2, bufA, stream, zero_page, fuse23 ? c3_pairs : nullptr, blk.conv3.kseg, nullptr,
                             fuse12 ? bufB : nullptr)))
            return rc;
        ActView a2{t2, blk.conv2.cout, blk.conv2.npad, true};
        if (blk.stride == 2) {
            const int64_t threads = (int64_t)B * OH * OW * a2.C;
            hipLaunchKernelGGL((avgpool2_ld_kernel<true>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, a2.p, B, H, W,
                               a2.C, a2.ld, a2.C, t1);
            LAUNCH_CHECK();
            a2 = ActView{t1, a2.C, a2.C, false};
        }
        const int res_mode = (dst == x.p && x.dirty) ? 2 : 1;
        // the block output as the next block's conv1 operand, when there is a next block and the layouts agree
        _Float16 *xp_next = (c3_pairs == bufA) ? bufB : bufA;
        bool emit = false;
        if (bi + 1 < cfg->n_blocks) {
            const mpreid_rn50_conv_split &n1 = w->blocks[bi + 1].conv1;
            emit = ((int64_t)B * OH * OW) % 256 == 0 && n1.taps == 1 && n1.kseg == blk.conv3.npad && n1.cin == blk.conv3.cout &&
                   blk.conv3.npad == blk.conv3.cout;
        }
        if ((rc = conv_split(blk.conv3, a2, B, OH, OW, res_mode, dst, c3_pairs, stream, nullptr, emit ? xp_next : nullptr, blk.conv3.npad,
                             fuse23 ? c3_pairs : nullptr)))
            return rc;
        have_xp = emit;
        if (emit) {   // the roles of the two buffers swap when the next block's input pairs sit in the other one
            bufB = (xp_next == bufA) ? bufB : bufA;
            bufA = xp_next;
        }
        x = ActView{dst, blk.conv3.cout, blk.conv3.npad, true};
        H = OH;
        W = OW;
    }
    ARG_CHECK(H * W == v.f.S && x.C == v.f.E && x.ld == v.f.E && x.dirty);   // (the last block's ReLU: applied by the tokens kernel on read)

    // ---- attention pool: k / v projections of the B * T tokens as pair GEMMs, the rest on the fp32 path ----
    float *mean = (float *)(base + v.f.mean), *tok = (float *)(base + v.f.tok), *tok0 = (float *)(base + v.f.tok0);
    float *q = (float *)(base + v.f.q), *k = (float *)(base + v.f.k), *vv = (float *)(base + v.f.v);
    float *att = (float *)(base + v.f.att), *proj = (float *)(base + v.f.proj);
    hipLaunchKernelGGL(tokens_f32_kernel<true>, dim3(B, (v.f.E + 255) / 256), dim3(256), 0, stream, x.p, wf.pos_emb, v.f.S, v.f.E, mean, tok);
    hipLaunchKernelGGL(gather_tok0_f32_kernel, dim3(B), dim3(256), 0, stream, tok, v.f.T, v.f.E, tok0);
    LAUNCH_CHECK();
    if ((rc = mpreid_gemm_f32_linear(tok0, wf.q_w, B, v.f.E, v.f.E, wf.q_b, q, v.f.E, F32_LIN, stream))) return rc;
    {
        ActView tk{tok, v.f.E, v.f.E, false};
        ARG_CHECK(w->k.npad == v.f.E && w->v.npad == v.f.E);
        // one pack serves both projections: conv_split packs per call, so pack once by hand and launch the two GEMMs
        if ((rc = conv_split(w->k, tk, B * v.f.T, 1, 1, 0, k, pairs, stream))) return rc;
        GemmArgs g{};
        g.A = pairs; g.W = (const _Float16 *)w->v.w; g.M = (int)pad_rows((int64_t)B * v.f.T); g.N = w->v.npad; g.K = 2 * w->v.kseg;
        g.kseg = w->v.kseg; g.oscale = w->v.oscale; g.out = vv; g.ldo = w->v.npad; g.bias = w->v.bias;
        ARG_CHECK(w->v.kseg == w->k.kseg);
        if ((rc = launch_gemm_f16(g, GE_S_BIAS_F32, stream))) return rc;
    }
    hipLaunchKernelGGL(pool_attend_f32_kernel, dim3((unsigned)(B * cfg->heads)), dim3(64), (size_t)v.f.T * 4, stream, q, k, vv, v.f.T,
                       v.f.E, cfg->heads, att);
    LAUNCH_CHECK();
    if ((rc = mpreid_gemm_f32_linear(att, wf.c_w, B, cfg->out_dim, v.f.E, wf.c_b, proj, cfg->out_dim, F32_LIN, stream))) return rc;
    hipLaunchKernelGGL(head_f32_kernel, dim3(B), dim3(256), 0, stream, mean, proj, v.f.E, cfg->out_dim, wf.bn_scale, wf.bn_shift, out);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int mpreid_rn50_forward_split(const mpreid_rn50_cfg *cfg, const mpreid_rn50_weights_split *w, const float *img, int B,
                                         float *out, void *ws, size_t ws_bytes, mpreid_stream_t stream_) {
    ARG_CHECK(img);
    return rn50_forward_split_impl(cfg, w, stem_in(img, nullptr, nullptr, nullptr), B, out, ws, ws_bytes, stream_);
}

extern "C" int mpreid_rn50_forward_split_u8(const mpreid_rn50_cfg *cfg, const mpreid_rn50_weights_split *w, const uint8_t *img_hwc,
                                            const float *mean, const float *stdv, int B, float *out, void *ws, size_t ws_bytes,
                                            mpreid_stream_t stream_) {
    ARG_CHECK(img_hwc && mean && stdv);
    return rn50_forward_split_impl(cfg, w, stem_in(nullptr, img_hwc, mean, stdv), B, out, ws, ws_bytes, stream_);
}

extern "C" int mpreid_rn50_forward_split_view(const mpreid_rn50_cfg *cfg, const mpreid_rn50_weights_split *w, const float *img_f32,
                                              const uint8_t *img_hwc, const float *mean, const float *stdv, int view, int B,
                                              float *out, void *ws, size_t ws_bytes, mpreid_stream_t stream_) {
    ARG_CHECK((img_f32 != nullptr) != (img_hwc != nullptr) && (!img_hwc || (mean && stdv)) && view >= 0 && view <= 3);
    return rn50_forward_split_impl(cfg, w, stem_in(img_f32, img_hwc, mean, stdv, view), B, out, ws, ws_bytes, stream_);
}
